"""Batch engine: thin object wrappers over the C ABI (include/oval_hip.h).

Device memory and streams come from PyTorch (plumbing only): a body is a 1-D
uint8 CUDA tensor, results a (n_obj, 4) int32/uint32-viewed tensor, counters a
(12,) int64 tensor that torch.distributed can all-reduce over RCCL.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import DESC_DTYPE, NUM_COUNTERS, RESULT_DTYPE, UNIT, OvalError, check


def device_count() -> int:
    n = C.c_int(0)
    rc = _lib.lib().oval_device_count(C.byref(n))
    return n.value if rc == 0 else 0


SMALL_OBJECT_BYTES_DEFAULT = (3072, 1280)      # OVAL_SMALL_FILL / VALIDATE_BYTES_DEFAULT of include/oval_hip.h


def set_small_object_bytes(fill_bytes: int, validate_bytes: int | None = None) -> None:
    """Layouts created or reset from now on whose objects average fewer than
    `fill_bytes` / `validate_bytes` take the fill / validate kernel for objects of
    a few units (0 = never, 2**64 - 1 = every layout; one argument sets both).  The
    reference's smallest object is one 256-byte unit
    (internal/pattern/pattern.go:23-34).  Results are identical either way."""
    check(_lib.lib().oval_set_small_object_bytes(fill_bytes, fill_bytes if validate_bytes is None else validate_bytes))


BIG_OBJECT_BYTES_DEFAULT = 1 << 20             # OVAL_BIG_VALIDATE_BYTES_DEFAULT of include/oval_hip.h


def set_big_object_bytes(validate_bytes: int) -> None:
    """Layouts created or reset from now on whose objects average at least
    `validate_bytes` take the validate kernel for multipart-sized bodies (one load
    per thread, one lookup per workgroup of 1024 threads); 0 = every layout,
    2**64 - 1 = never.  Results are identical either way."""
    check(_lib.lib().oval_set_big_object_bytes(validate_bytes))


class Context:
    """One device + one stream (oval_ctx)."""

    def __init__(self, device: int = 0, stream_ptr: int | None = None):
        self._h = C.c_void_p()
        check(_lib.lib().oval_ctx_create(device, C.c_void_p(stream_ptr or 0), C.byref(self._h)))
        self.device = device

    @classmethod
    def on_torch_stream(cls, device: int = 0):
        """Context whose stream is torch's current stream on `device`, so that
        torch ops and engine launches are ordered with each other."""
        import torch

        with torch.cuda.device(device):
            cur = torch.cuda.current_stream(device)
            keep = None
            if not cur.cuda_stream:
                # torch is on the null stream, which the ABI reads as "create
                # your own": give torch a real stream and share that one
                keep = torch.cuda.Stream(device)
                torch.cuda.set_stream(keep)
                cur = keep
            ctx = cls(device, cur.cuda_stream)
            ctx._torch_stream = cur      # keeps the stream alive as long as the context
        return ctx

    @property
    def handle(self):
        return self._h

    def set_tuning(self, fill_threads: int = 0, validate_threads: int = 0) -> None:
        """Workgroup sizes of the two kernels (128/256/512/1024; 0 = default)."""
        check(_lib.lib().oval_ctx_set_tuning(self._h, fill_threads, validate_threads))

    def sync(self) -> None:
        check(_lib.lib().oval_ctx_sync(self._h))

    def stream_ptr(self) -> int:
        p = C.c_void_p()
        check(_lib.lib().oval_ctx_stream(self._h, C.byref(p)))
        return p.value or 0

    def close(self) -> None:
        if self._h:
            _lib.lib().oval_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- one-shot helpers for host-resident bodies ---------------------------
    def fill_batch_host(self, descs: np.ndarray, body: np.ndarray) -> None:
        _check_descs(descs)
        assert body.dtype == np.uint8 and body.flags.c_contiguous
        check(_lib.lib().oval_fill_batch_host(self._h, descs.ctypes.data, len(descs), body.ctypes.data, body.size))

    def validate_batch_host(self, descs: np.ndarray, body: np.ndarray):
        _check_descs(descs)
        assert body.dtype == np.uint8 and body.flags.c_contiguous
        res = np.empty(len(descs), dtype=RESULT_DTYPE)
        cnt = np.zeros(NUM_COUNTERS, dtype=np.uint64)
        check(_lib.lib().oval_validate_batch_host(self._h, descs.ctypes.data, len(descs), body.ctypes.data,
                                                  body.size, res.ctypes.data, cnt.ctypes.data))
        return res, cnt


def _check_descs(descs: np.ndarray) -> None:
    if descs.dtype != DESC_DTYPE or not descs.flags.c_contiguous:
        raise OvalError("descriptors must be a C-contiguous array of DESC_DTYPE")


class Layout:
    """Device-resident descriptor table + chunk index (oval_layout)."""

    def __init__(self, ctx: Context, descs: np.ndarray):
        _check_descs(descs)
        self.ctx = ctx
        self._h = C.c_void_p()
        check(_lib.lib().oval_layout_create(ctx.handle, descs.ctypes.data, len(descs), C.byref(self._h)))
        self._read_info()

    def _read_info(self):
        n, u, s = C.c_uint64(), C.c_uint64(), C.c_uint64()
        check(_lib.lib().oval_layout_info(self._h, C.byref(n), C.byref(u), C.byref(s)))
        self.n_obj, self.total_units, self.span_bytes = n.value, u.value, s.value

    def launches(self):
        """(fill, validate) kernel launches one pass over this layout takes."""
        f, v = C.c_uint32(), C.c_uint32()
        check(_lib.lib().oval_layout_launches(self._h, C.byref(f), C.byref(v)))
        return f.value, v.value

    def reset(self, descs: np.ndarray, ahead: bool = False) -> None:
        """New descriptors for this layout; `ahead`: uploaded and indexed on the
        context's second stream, behind this layout's own last pass only
        (oval_layout_reset_ahead)."""
        _check_descs(descs)
        fn = _lib.lib().oval_layout_reset_ahead if ahead else _lib.lib().oval_layout_reset
        check(fn(self.ctx.handle, self._h, descs.ctypes.data, len(descs)))
        self._read_info()

    def reset_compact(self, segs: np.ndarray, recs: np.ndarray, ahead: bool = False) -> None:
        """New contents from 12-byte records and a segment table
        (oval_layout_reset_compact): the descriptors are written on the device;
        `ahead` as in reset()."""
        if segs.dtype != _lib.SEGMENT_DTYPE or recs.dtype != _lib.REC_DTYPE:
            raise TypeError("segs / recs must have SEGMENT_DTYPE / REC_DTYPE")
        if not (segs.flags["C_CONTIGUOUS"] and recs.flags["C_CONTIGUOUS"]):
            raise ValueError("segs / recs must be contiguous")
        check(_lib.lib().oval_layout_reset_compact(self.ctx.handle, self._h, segs.ctypes.data, len(segs),
                                                   recs.ctypes.data, len(recs), 1 if ahead else 0))
        self._read_info()

    def read_descs(self, first: int = 0, n: int | None = None) -> np.ndarray:
        """Descriptors [first, first + n) of the device table, copied back (waits)."""
        n = self.n_obj - first if n is None else n
        out = np.zeros(n, dtype=_lib.DESC_DTYPE)
        check(_lib.lib().oval_layout_read_descs(self.ctx.handle, self._h, first, n, out.ctypes.data))
        return out

    def close(self) -> None:
        if self._h:
            _lib.lib().oval_layout_destroy(self.ctx.handle, self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- the two kernels ------------------------------------------------------
    def fill(self, body) -> None:
        """body: 1-D uint8 CUDA tensor of at least span_bytes bytes."""
        _check_body(body, self.ctx.device)
        check(_lib.lib().oval_fill(self.ctx.handle, self._h, body.data_ptr(), body.numel()))

    def validate(self, body, results=None, counters=None):
        """Returns (results, counters) tensors; counters are ADDED to."""
        import torch

        _check_body(body, self.ctx.device)
        dev = body.device
        if results is None:
            results = torch.empty((max(self.n_obj, 1), 4), dtype=torch.int32, device=dev)
        if counters is None:
            counters = torch.zeros(NUM_COUNTERS, dtype=torch.int64, device=dev)
        assert results.is_contiguous() and results.numel() * results.element_size() >= 16 * self.n_obj
        assert counters.is_contiguous() and counters.numel() >= NUM_COUNTERS and counters.element_size() == 8
        check(_lib.lib().oval_validate(self.ctx.handle, self._h, body.data_ptr(), body.numel(),
                                       results.data_ptr(), counters.data_ptr()))
        return results, counters

    def read_unit(self, body, byte_offset: int) -> bytes:
        out = (C.c_uint8 * UNIT)()
        check(_lib.lib().oval_read_unit(self.ctx.handle, body.data_ptr(), byte_offset, out))
        return bytes(out)


def _check_body(body, device: int) -> None:
    if not (body.is_cuda and body.dtype.itemsize == 1 and body.is_contiguous()):
        raise OvalError("body must be a contiguous 1-byte-element CUDA tensor")
    if body.device.index != device:
        raise OvalError(f"body is on cuda:{body.device.index}, context on cuda:{device}")


def results_to_numpy(results, n_obj: int) -> np.ndarray:
    """(n_obj,4) int32 CUDA tensor -> structured RESULT_DTYPE array on the host."""
    arr = results[:n_obj].cpu().numpy().view(np.uint32).reshape(-1, 4)
    out = np.empty(n_obj, dtype=RESULT_DTYPE)
    out["first_bad_fields"] = arr[:, 0]
    out["first_bad_unit"] = arr[:, 1]
    out["bad_units"] = arr[:, 2]
    out["strict_first_bad_unit"] = arr[:, 3]
    return out


def gather_first_bad(ctx: "Context", results, n_obj: int, table_slice) -> None:
    """table_slice[i] = first_bad_unit of results[i] (uint32 in an int32 CUDA tensor), on the context's stream."""
    if table_slice.numel() < n_obj:
        raise ValueError("table slice shorter than the result array")
    check(_lib.lib().oval_gather_first_bad(ctx.handle, results.data_ptr(), n_obj, table_slice.data_ptr()))


class Comm:
    """RCCL communicator behind oval_allreduce_counters: one rank per GPU, the
    batch counters are the only thing summed over ranks."""

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        check(_lib.lib().oval_comm_unique_id(buf))
        return buf.raw

    def __init__(self, ctx: "Context", n_ranks: int, rank: int, uid: bytes):
        if len(uid) != 128:
            raise ValueError("the unique ID is 128 bytes")
        self.ctx = ctx
        self._h = C.c_void_p()
        check(_lib.lib().oval_comm_create(ctx.handle, n_ranks, rank, uid, C.byref(self._h)))

    def info(self):
        """(ranks in the communicator, this rank), as RCCL reports them."""
        n, me = C.c_int(), C.c_int()
        check(_lib.lib().oval_comm_info(self._h, C.byref(n), C.byref(me)))
        return n.value, me.value

    def all_reduce_counters(self, counters) -> None:
        """counters: int64 CUDA tensor, summed over all ranks in place on the context's stream."""
        check(_lib.lib().oval_allreduce_counters(self.ctx.handle, self._h, counters.data_ptr(), counters.numel()))

    def all_reduce_first_bad(self, table) -> None:
        """table: int32 CUDA tensor holding uint32 first-bad units (0xFFFFFFFF = none), one per object of
        the whole job; the minimum over all ranks, in place on the context's stream."""
        check(_lib.lib().oval_allreduce_first_bad(self.ctx.handle, self._h, table.data_ptr(), table.numel()))

    def close(self) -> None:
        if self._h:
            _lib.lib().oval_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Timer:
    """HIP events on the context's stream (oval_timer)."""

    def __init__(self, ctx: Context):
        self.ctx = ctx
        self._h = C.c_void_p()
        check(_lib.lib().oval_timer_create(ctx.handle, C.byref(self._h)))

    def start(self):
        check(_lib.lib().oval_timer_start(self.ctx.handle, self._h))

    def stop(self):
        check(_lib.lib().oval_timer_stop(self.ctx.handle, self._h))

    def elapsed_ms(self) -> float:
        ms = C.c_float()
        check(_lib.lib().oval_timer_elapsed_ms(self._h, C.byref(ms)))
        return ms.value

    def __del__(self):
        try:
            if self._h:
                _lib.lib().oval_timer_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

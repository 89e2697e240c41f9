"""Mirror of the reference's internal/pattern package, bound to the C++ host
layer (oval_amd/csrc/oval_pattern.cpp) which drives the HIP kernels.

Go functions that return `error` raise PatternError here, whose text is what
err.Error() returns in the reference; `None`/normal return stands for nil.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import OvalError
from .object import Object

dataUnitSize = 256  # pattern.go:18
_ERR_CAP = 8192


class PatternError(Exception):
    """A Go `error` value of internal/pattern."""


def _b(s) -> bytes:
    return s.encode("utf-8") if isinstance(s, str) else bytes(s)


def _raise(rc: int, err) -> None:
    if rc == 0:
        return
    text = err.value.decode("utf-8", "surrogateescape")
    if rc < 0:
        raise OvalError(text or _lib.last_error())
    raise PatternError(text)


def init(device: int = 0) -> None:
    _lib.check(_lib.lib().oval_pattern_init(device))


def init_engines(devices) -> None:
    """One engine per entry of `devices` (a GPU may be named more than once);
    calls are routed by worker ID modulo the engine count."""
    arr = (C.c_int * len(devices))(*devices)
    _lib.check(_lib.lib().oval_pattern_init_engines(arr, len(devices)))


def engine_count() -> int:
    return _lib.lib().oval_pattern_engine_count()


def set_device_timeout(ms: int) -> None:
    """How long a call waits for the device before it fails (default 30000 ms; 0 = no limit)."""
    _lib.lib().oval_pattern_set_device_timeout(ms)


def set_caller_spin(spin_ns: int = -1, max_spinners: int = -1) -> None:
    _lib.lib().oval_pattern_set_caller_spin(spin_ns, max_spinners)


def test_lose_flags(n: int) -> None:
    """Test hook: the next n small launches never report their end (a hung device)."""
    _lib.lib().oval_pattern_test_lose_flags(n)


test_lose_flags.__test__ = False          # not a pytest test


def quarantined_slots() -> int:
    return _lib.lib().oval_pattern_quarantined_slots()


def shutdown() -> None:
    _lib.lib().oval_pattern_shutdown()


def set_clock(unix_micro: int) -> None:
    """Pins time.Now().UnixMicro() (pattern.go:103-104); < 0 = wall clock."""
    _lib.lib().oval_pattern_set_clock(unix_micro)


def seed(s: int) -> None:
    _lib.lib().oval_pattern_seed(s)


def DecideSize(minSize: int, maxSize: int, y: float | None = None) -> int:
    """pattern.go:22-71; `y` optionally supplies rand.Float64()."""
    size = C.c_int64()
    err = C.create_string_buffer(256)
    if y is None:
        rc = _lib.lib().oval_pattern_decide_size(minSize, maxSize, C.byref(size), err, 256)
    else:
        rc = _lib.lib().oval_pattern_decide_size_y(minSize, maxSize, y, C.byref(size), err, 256)
    _raise(rc, err)
    return size.value


def Generate(dataSize: int, workerID: int, bucketName, obj: Object) -> bytes:
    """pattern.go:73-92"""
    b, k = _b(bucketName), _b(obj.Key)
    out = np.empty(max(dataSize, 1), dtype=np.uint8)
    err = C.create_string_buffer(_ERR_CAP)
    rc = _lib.lib().oval_pattern_generate(dataSize, workerID, b, len(b), k, len(k), obj.WriteCount,
                                          out.ctypes.data, err, _ERR_CAP)
    _raise(rc, err)
    return out[:dataSize].tobytes()


class GeneratedParts:
    """Generate whose result leaves in parts (the reference's multipartUpload,
    internal/s3client/s3client.go:164-236, slices body[:partSize] at :185 and
    body = body[partSize:] at :204): one fill into pinned memory owned by the
    library, part views handed out of it and released one by one."""

    def __init__(self, handle):
        self._h = handle

    def __len__(self) -> int:
        return _lib.lib().oval_parts_count(self._h) if self._h else 0

    def _handle(self):
        if self._h is None:
            raise OvalError("this generated body has been closed: its buffer is back in the library's pool")
        return self._h

    def part(self, i: int) -> np.ndarray:
        """A view (not a copy) of part i, straight into the library's pooled pinned
        buffer: valid until release(i) / close() and not kept alive by numpy.  The
        buffer is recycled as soon as the last part has been released, so take a
        copy (`.copy()`) of anything that must outlive its release."""
        self._handle()
        data, n = C.c_void_p(), C.c_size_t()
        err = C.create_string_buffer(_ERR_CAP)
        rc = _lib.lib().oval_parts_get(self._h, i, C.byref(data), C.byref(n), err, _ERR_CAP)
        _raise(rc, err)
        if n.value == 0:
            return np.empty(0, dtype=np.uint8)
        return np.ctypeslib.as_array(C.cast(data, C.POINTER(C.c_uint8)), shape=(n.value,))

    def release(self, i: int) -> None:
        _lib.check(_lib.lib().oval_parts_release(self._handle(), i))

    def close(self) -> None:
        if self._h:
            _lib.lib().oval_parts_close(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        self.close()


def GenerateParts(dataSize: int, workerID: int, bucketName, obj: Object, partSize: int) -> GeneratedParts:
    """pattern.go:73-92 for a body that is uploaded in parts of partSize bytes."""
    b, k = _b(bucketName), _b(obj.Key)
    h = C.c_void_p()
    err = C.create_string_buffer(_ERR_CAP)
    rc = _lib.lib().oval_pattern_generate_parts(dataSize, workerID, b, len(b), k, len(k), obj.WriteCount, partSize,
                                                C.byref(h), err, _ERR_CAP)
    _raise(rc, err)
    return GeneratedParts(h)


def parts_pool_stats():
    n, nbytes = C.c_size_t(), C.c_size_t()
    _lib.lib().oval_parts_pool_stats(C.byref(n), C.byref(nbytes))
    return n.value, nbytes.value


def generateDataUnit(unitCount: int, workerID: int, bucketName, obj: Object) -> bytes:
    """pattern.go:94-116"""
    b, k = _b(bucketName), _b(obj.Key)
    out = np.empty(dataUnitSize, dtype=np.uint8)
    err = C.create_string_buffer(_ERR_CAP)
    rc = _lib.lib().oval_pattern_generate_data_unit(unitCount, workerID, b, len(b), k, len(k), obj.WriteCount,
                                                    out.ctypes.data, err, _ERR_CAP)
    _raise(rc, err)
    return out.tobytes()


def Valid(workerID: int, expectedBucketName, obj: Object, reader) -> None:
    """pattern.go:118-134; `reader` is bytes-like (the whole body the io.Reader
    would deliver)."""
    b, k = _b(expectedBucketName), _b(obj.Key)
    body = np.frombuffer(_b(reader), dtype=np.uint8) if not isinstance(reader, np.ndarray) else reader
    body = np.ascontiguousarray(body, dtype=np.uint8)
    err = C.create_string_buffer(_ERR_CAP)
    ptr = body.ctypes.data if body.size else None
    rc = _lib.lib().oval_pattern_valid(workerID, b, len(b), k, len(k), obj.Size, obj.WriteCount, ptr, body.size,
                                       err, _ERR_CAP)
    _raise(rc, err)


_READ_FN = C.CFUNCTYPE(C.c_size_t, C.c_void_p, C.POINTER(C.c_uint8), C.c_size_t)


def ValidStream(workerID: int, expectedBucketName, obj: Object, reader) -> None:
    """pattern.go:118-134 with a file-like reader (`reader.read(n)` returns up
    to n bytes, b"" at EOF); the body is validated chunk by chunk."""
    b, k = _b(expectedBucketName), _b(obj.Key)

    def cb(_user, p, n):
        data = reader.read(n)
        if not data:
            return 0
        C.memmove(p, data, len(data))
        return len(data)

    fn = _READ_FN(cb)
    err = C.create_string_buffer(_ERR_CAP)
    rc = _lib.lib().oval_pattern_valid_stream(workerID, b, len(b), k, len(k), obj.Size, obj.WriteCount,
                                              C.cast(fn, C.c_void_p), None, err, _ERR_CAP)
    _raise(rc, err)


def set_stream_chunk(nbytes: int) -> None:
    _lib.lib().oval_pattern_set_stream_chunk(nbytes)


def validDataUnit(unitCount: int, workerID: int, expectedBucketName, obj: Object, data) -> None:
    """pattern.go:136-180"""
    b, k, d = _b(expectedBucketName), _b(obj.Key), _b(data)
    if len(d) != dataUnitSize:
        raise ValueError("a data unit is 256 bytes")
    err = C.create_string_buffer(_ERR_CAP)
    arr = np.frombuffer(d, dtype=np.uint8)
    rc = _lib.lib().oval_pattern_valid_data_unit(unitCount, workerID, b, len(b), k, len(k), obj.WriteCount,
                                                 arr.ctypes.data, err, _ERR_CAP)
    _raise(rc, err)


class PendingCall:
    """Generate / Valid in two steps: begin returns at once, `poll()` returns
    False while the device works and True once the call is over, after which
    `result()` is the bytes (Generate) / None (Valid) or raises what the blocking
    call would have raised."""

    def __init__(self, handle, keep, out=None, size=0):
        self._h, self._keep, self._out, self._size = handle, keep, out, size
        self._rc, self._text = None, None

    def poll(self) -> bool:
        if self._rc is None:
            err = C.create_string_buffer(_ERR_CAP)
            rc = _lib.lib().oval_pattern_poll(self._h, err, _ERR_CAP)
            if rc == 2:
                return False
            self._rc, self._text, self._h = rc, err, None
        return True

    def result(self):
        while not self.poll():
            pass
        _raise(self._rc, self._text)
        return self._out[:self._size].tobytes() if self._out is not None else None

    def __del__(self):
        # a call that was begun must be collected: its staging slot is held until then
        try:
            while self._rc is None and not self.poll():
                pass
        except Exception:
            pass


def GenerateBegin(dataSize: int, workerID: int, bucketName, obj: Object) -> PendingCall:
    b, k = _b(bucketName), _b(obj.Key)
    out = np.empty(max(dataSize, 1), dtype=np.uint8)
    h = C.c_void_p()
    _lib.check(_lib.lib().oval_pattern_generate_begin(dataSize, workerID, b, len(b), k, len(k), obj.WriteCount,
                                                      out.ctypes.data, C.byref(h)))
    return PendingCall(h, (b, k), out, dataSize)


def ValidBegin(workerID: int, expectedBucketName, obj: Object, body) -> PendingCall:
    b, k = _b(expectedBucketName), _b(obj.Key)
    arr = np.ascontiguousarray(np.frombuffer(_b(body), dtype=np.uint8) if not isinstance(body, np.ndarray) else body,
                               dtype=np.uint8)
    h = C.c_void_p()
    _lib.check(_lib.lib().oval_pattern_valid_begin(workerID, b, len(b), k, len(k), obj.Size, obj.WriteCount,
                                                   arr.ctypes.data if arr.size else None, arr.size, C.byref(h)))
    return PendingCall(h, (b, k, arr))


def set_batching(enable: bool, window_us: int = 0, max_bytes: int = 64 << 20) -> None:
    """Coalesce concurrent Generate / Valid calls from different threads into
    shared device launches (results are unchanged)."""
    _lib.lib().oval_pattern_set_batching(1 if enable else 0, window_us, max_bytes)


def set_direct_callers(callers: int) -> None:
    """While batching is on, a call that finds at most `callers` calls under way
    launches on its own (default: the host's cores, at most 16; 0 = always through the micro-batcher)."""
    _lib.lib().oval_pattern_set_direct_callers(callers)


def set_lanes(lanes: int = 0, blocking_sync: int = -1) -> None:
    """Batches in flight at once / lane threads sleep instead of polling."""
    _lib.lib().oval_pattern_set_lanes(lanes, blocking_sync)


def batching_stats():
    b, j = C.c_uint64(), C.c_uint64()
    _lib.lib().oval_pattern_batching_stats(C.byref(b), C.byref(j))
    return b.value, j.value


def _name_arrays(names):
    bs = [_b(x) for x in names]
    ptrs = (C.c_char_p * len(bs))(*bs)
    lens = (C.c_size_t * len(bs))(*[len(x) for x in bs])
    return bs, ptrs, lens


def _errors(n, rcs, errs, stride):
    out = []
    for i in range(n):
        if rcs[i] == 0:
            out.append(None)
            continue
        text = errs.raw[i * stride:(i + 1) * stride].split(b"\0", 1)[0].decode("utf-8", "surrogateescape")
        out.append(OvalError(text) if rcs[i] < 0 else PatternError(text))
    return out


def GenerateBatch(items):
    """items: sequence of (dataSize, workerID, bucketName, Object).  One device
    launch for all of them.  Returns a list with, per item, the generated bytes
    or the PatternError that Generate would have returned."""
    n = len(items)
    if n == 0:
        return []
    sizes = (C.c_int64 * n)(*[it[0] for it in items])
    wids = (C.c_int64 * n)(*[it[1] for it in items])
    _bk, bptr, blen = _name_arrays([it[2] for it in items])
    _kk, kptr, klen = _name_arrays([it[3].Key for it in items])
    wcs = (C.c_int64 * n)(*[it[3].WriteCount for it in items])
    outs = [np.empty(max(it[0], 1), dtype=np.uint8) for it in items]
    optr = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
    rcs = (C.c_int * n)()
    errs = C.create_string_buffer(n * 512)
    _lib.lib().oval_pattern_generate_batch(n, sizes, wids, bptr, blen, kptr, klen, wcs, optr, rcs, errs, 512)
    es = _errors(n, rcs, errs, 512)
    return [es[i] if es[i] is not None else outs[i][:items[i][0]].tobytes() for i in range(n)]


def ValidBatch(items):
    """items: sequence of (workerID, expectedBucketName, Object, body bytes).
    Returns a list of None (nil) or the PatternError Valid would have returned."""
    n = len(items)
    if n == 0:
        return []
    wids = (C.c_int64 * n)(*[it[0] for it in items])
    _bk, bptr, blen = _name_arrays([it[1] for it in items])
    _kk, kptr, klen = _name_arrays([it[2].Key for it in items])
    sizes = (C.c_int64 * n)(*[it[2].Size for it in items])
    wcs = (C.c_int64 * n)(*[it[2].WriteCount for it in items])
    bodies = [np.frombuffer(_b(it[3]), dtype=np.uint8) if not isinstance(it[3], np.ndarray)
              else np.ascontiguousarray(it[3], dtype=np.uint8) for it in items]
    bp = (C.c_void_p * n)(*[(b.ctypes.data if b.size else None) for b in bodies])
    bl = (C.c_size_t * n)(*[b.size for b in bodies])
    rcs = (C.c_int * n)()
    errs = C.create_string_buffer(n * _ERR_CAP)
    _lib.lib().oval_pattern_valid_batch(n, wids, bptr, blen, kptr, klen, sizes, wcs, bp, bl, rcs, errs, _ERR_CAP)
    return _errors(n, rcs, errs, _ERR_CAP)


def hexDump(data) -> str:
    d = _b(data)
    cap = 80 * (len(d) // 16 + 2)
    out = C.create_string_buffer(cap)
    arr = np.frombuffer(d, dtype=np.uint8)
    n = _lib.lib().oval_pattern_hex_dump(arr.ctypes.data if len(d) else None, len(d), out, cap)
    return out.raw[:n].decode("latin-1")


def dump(data) -> str:
    """pattern.go:182-206"""
    d = _b(data) if not isinstance(data, str) else data.encode("latin-1")
    cap = len(d) + 1024
    out = C.create_string_buffer(cap)
    n = _lib.lib().oval_pattern_dump(d, len(d), out, cap)
    return out.raw[:n].decode("latin-1")

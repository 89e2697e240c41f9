"""ctypes front-end of the CPU oracle (oracle/oval_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under oval_amd/ imports this module.

The functions mirror the reference's Go signatures
(internal/pattern/pattern.go:22,73,94,118,136,182) with the wall clock and the
random sample made explicit arguments.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboval_oracle.so")

UNIT = 256
NO_BAD = 0xFFFFFFFF

# Same layout as include/oval_hip.h's oval_obj_desc / oval_obj_result.
DESC_DTYPE = np.dtype(
    [
        ("bucket", "u1", (16,)),
        ("key", "u1", (12,)),
        ("write_count", "<u4"),
        ("first_unit", "<u4"),
        ("worker_id", "<u4"),
        ("unix_micro", "<u8"),
        ("body_offset", "<u8"),
        ("n_units", "<u4"),
        ("bucket_len", "u1"),
        ("key_len", "u1"),
        ("flags", "<u2"),
    ],
    align=False,
)
RESULT_DTYPE = np.dtype(
    [
        ("first_bad_fields", "<u4"),
        ("first_bad_unit", "<u4"),
        ("bad_units", "<u4"),
        ("strict_first_bad_unit", "<u4"),
    ]
)
NUM_COUNTERS = 12
assert DESC_DTYPE.itemsize == 64 and RESULT_DTYPE.itemsize == 16


def build() -> str:
    """Compile the oracle if the shared object is missing or stale."""
    src = os.path.join(_HERE, "oval_oracle.c")
    hdr = os.path.join(_HERE, "oval_oracle.h")
    if (not os.path.exists(_LIB_PATH)) or os.path.getmtime(_LIB_PATH) < max(
        os.path.getmtime(src), os.path.getmtime(hdr)
    ):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboval_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        u8p = C.POINTER(C.c_uint8)
        L.oracle_rune_count.restype = C.c_size_t
        L.oracle_rune_count.argtypes = [C.c_char_p, C.c_size_t]
        L.oracle_trim_space.restype = None
        L.oracle_trim_space.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        L.oracle_pad_field.restype = C.c_size_t
        L.oracle_pad_field.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_char_p]
        L.oracle_generate_key.restype = C.c_size_t
        L.oracle_generate_key.argtypes = [C.c_int64, C.c_char_p]
        L.oracle_generate_unit.restype = C.c_size_t
        L.oracle_generate_unit.argtypes = [C.c_int64, C.c_int64, C.c_char_p, C.c_size_t, C.c_char_p,
                                           C.c_size_t, C.c_int64, C.c_uint64, C.c_char_p]
        L.oracle_generate.restype = C.c_int
        L.oracle_generate.argtypes = [C.c_int64, C.c_int64, C.c_char_p, C.c_size_t, C.c_char_p,
                                      C.c_size_t, C.c_int64, C.c_uint64, C.c_void_p, C.c_char_p,
                                      C.c_size_t]
        L.oracle_valid_unit.restype = C.c_int
        L.oracle_valid_unit.argtypes = [C.c_int64, C.c_int64, C.c_char_p, C.c_size_t, C.c_char_p,
                                        C.c_size_t, C.c_int64, C.c_char_p, C.c_char_p, C.c_size_t]
        L.oracle_valid.restype = C.c_int
        L.oracle_valid.argtypes = [C.c_int64, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t,
                                   C.c_int64, C.c_int64, C.c_void_p, C.c_size_t,
                                   C.POINTER(C.c_int64), C.c_char_p, C.c_size_t]
        L.oracle_hex_dump.restype = C.c_size_t
        L.oracle_hex_dump.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
        L.oracle_dump.restype = C.c_size_t
        L.oracle_dump.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
        L.oracle_decide_size.restype = C.c_int
        L.oracle_decide_size.argtypes = [C.c_int64, C.c_int64, C.c_double, C.POINTER(C.c_int64),
                                         C.c_char_p, C.c_size_t]
        L.oracle_fill_batch.restype = None
        L.oracle_fill_batch.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
        L.oracle_validate_batch.restype = None
        L.oracle_validate_batch.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                            C.c_void_p, C.c_void_p]
        L.oracle_refshape_generate_valid.restype = C.c_uint64
        L.oracle_refshape_generate_valid.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
        _ = u8p
        _lib = L
    return _lib


def _b(s) -> bytes:
    return s.encode("utf-8") if isinstance(s, str) else bytes(s)


class OracleError(Exception):
    """Carries the text of a Go `error` value."""


# --------------------------------------------------------------------------
# scalar API (Go signatures)
# --------------------------------------------------------------------------
def rune_count(s) -> int:
    s = _b(s)
    return lib().oracle_rune_count(s, len(s))


def trim_space(s) -> bytes:
    s = _b(s)
    a, b = C.c_size_t(), C.c_size_t()
    lib().oracle_trim_space(s, len(s), C.byref(a), C.byref(b))
    return s[a.value:b.value]


def pad_field(s, width: int) -> bytes:
    s = _b(s)
    out = C.create_string_buffer(len(s) + width + 1)
    n = lib().oracle_pad_field(s, len(s), width, out)
    return out.raw[:n]


def generate_key(obj_id: int) -> str:
    out = C.create_string_buffer(32)
    n = lib().oracle_generate_key(obj_id, out)
    return out.raw[:n].decode()


def generate_data_unit(unit_count: int, worker_id: int, bucket, key, write_count: int,
                       unix_micro: int) -> bytes:
    """pattern.go:94 generateDataUnit (obj reduced to key + write count)."""
    bucket, key = _b(bucket), _b(key)
    out = C.create_string_buffer(len(bucket) + len(key) + 28 + 256 + 64)
    n = lib().oracle_generate_unit(unit_count, worker_id, bucket, len(bucket), key, len(key),
                                   write_count, unix_micro, out)
    return out.raw[:n]


def generate(data_size: int, worker_id: int, bucket, key, write_count: int,
             unix_micro: int) -> bytes:
    """pattern.go:73 Generate; raises OracleError with the reference's text."""
    bucket, key = _b(bucket), _b(key)
    out = np.empty(max(data_size, 1), dtype=np.uint8)
    err = C.create_string_buffer(256)
    rc = lib().oracle_generate(data_size, worker_id, bucket, len(bucket), key, len(key),
                               write_count, unix_micro, out.ctypes.data, err, 256)
    if rc != 0:
        raise OracleError(err.value.decode())
    return out[:data_size].tobytes()


def valid_data_unit(unit_count: int, worker_id: int, bucket, key, write_count: int, data):
    """pattern.go:136 validDataUnit -> (mask, error text or None)."""
    bucket, key, data = _b(bucket), _b(key), _b(data)
    assert len(data) == UNIT
    err = C.create_string_buffer(4096)
    m = lib().oracle_valid_unit(unit_count, worker_id, bucket, len(bucket), key, len(key),
                                write_count, data, err, 4096)
    return m, (err.value.decode("utf-8", "surrogateescape") if m else None)


def valid(worker_id: int, bucket, key, size: int, write_count: int, body):
    """pattern.go:118 Valid over an in-memory reader.

    Returns (kind, unit, text): kind 0 = nil, 1 = unit mismatch, 2 = short read.
    """
    bucket, key = _b(bucket), _b(key)
    if isinstance(body, np.ndarray):
        arr = np.ascontiguousarray(body, dtype=np.uint8)
    else:
        arr = np.frombuffer(_b(body), dtype=np.uint8)
    bad = C.c_int64(-1)
    err = C.create_string_buffer(4096)
    ptr = arr.ctypes.data if arr.size else None
    rc = lib().oracle_valid(worker_id, bucket, len(bucket), key, len(key), size, write_count,
                            ptr, arr.size, C.byref(bad), err, 4096)
    return rc, bad.value, (err.value.decode("utf-8", "surrogateescape") if rc else None)


def hex_dump(data) -> str:
    data = _b(data)
    cap = 80 * (len(data) // 16 + 2)
    out = C.create_string_buffer(cap)
    n = lib().oracle_hex_dump(data, len(data), out, cap)
    return out.raw[:n].decode("latin-1")


def dump(text) -> str:
    text = _b(text)
    cap = len(text) + 1024
    out = C.create_string_buffer(cap)
    n = lib().oracle_dump(text, len(text), out, cap)
    return out.raw[:n].decode("latin-1")


def decide_size(min_size: int, max_size: int, y: float) -> int:
    """pattern.go:22 DecideSize with rand.Float64() replaced by y."""
    size = C.c_int64()
    err = C.create_string_buffer(256)
    rc = lib().oracle_decide_size(min_size, max_size, y, C.byref(size), err, 256)
    if rc != 0:
        raise OracleError(err.value.decode())
    return size.value


# --------------------------------------------------------------------------
# batch API (descriptor arrays)
# --------------------------------------------------------------------------
def fill_batch(descs: np.ndarray, body: np.ndarray) -> None:
    assert descs.dtype == DESC_DTYPE and descs.flags.c_contiguous
    assert body.dtype == np.uint8 and body.flags.c_contiguous
    if len(descs):
        end = int((descs["body_offset"] + descs["n_units"].astype(np.uint64) * UNIT).max())
        assert end <= body.size, "body too small for the descriptors"
    lib().oracle_fill_batch(descs.ctypes.data, len(descs), body.ctypes.data)


def validate_batch(descs: np.ndarray, body: np.ndarray):
    assert descs.dtype == DESC_DTYPE and descs.flags.c_contiguous
    assert body.dtype == np.uint8 and body.flags.c_contiguous
    res = np.empty(len(descs), dtype=RESULT_DTYPE)
    counters = np.zeros(NUM_COUNTERS, dtype=np.uint64)
    lib().oracle_validate_batch(descs.ctypes.data, len(descs), body.ctypes.data, body.size,
                                res.ctypes.data, counters.ctypes.data)
    return res, counters


def refshape_generate_valid(descs: np.ndarray):
    """Timing variant shaped like pattern.go's per-unit work (CPU baseline only):
    Generate then Valid of every object.  Returns (units, bad units)."""
    assert descs.dtype == DESC_DTYPE and descs.flags.c_contiguous
    bad = C.c_uint64()
    units = lib().oracle_refshape_generate_valid(descs.ctypes.data, len(descs), C.byref(bad))
    return int(units), int(bad.value)

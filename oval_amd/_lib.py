"""Loader and ctypes signatures of liboval_hip.so (include/oval_hip.h,
include/oval_pattern.h).

There is no Python or CPU implementation of the pattern engine in this package:
if the shared library is missing, importing anything that needs it raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# OVAL_HIP_LIB names another build of the same library (A/B measurements of two builds on one box)
LIB_PATH = os.environ.get("OVAL_HIP_LIB") or os.path.join(_HERE, "liboval_hip.so")

UNIT = 256
# enum oval_status
(OVAL_OK, OVAL_E_INVALID, OVAL_E_HIP, OVAL_E_NOMEM, OVAL_E_LAYOUT, OVAL_E_FIELD, OVAL_E_RANGE, OVAL_E_NODEVICE,
 OVAL_E_INTERNAL) = (0, -1, -2, -3, -4, -5, -6, -7, -8)
NO_BAD_UNIT = 0xFFFFFFFF
NUM_COUNTERS = 12
(CNT_OBJECTS_CHECKED, CNT_OBJECTS_BAD, CNT_UNITS_CHECKED, CNT_UNITS_BAD, CNT_BAD_BUCKET, CNT_BAD_KEY,
 CNT_BAD_WRITECNT, CNT_BAD_OFFSET, CNT_BAD_WORKER, CNT_STRICT_UNITS_BAD) = range(10)
BAD_BUCKET, BAD_KEY, BAD_WRITECNT, BAD_OFFSET, BAD_WORKER, BAD_FILLER = 1, 2, 4, 8, 16, 32
DESC_BUCKET_SLOW, DESC_KEY_SLOW, DESC_WORKER_OOR, DESC_NO_FILL = 1, 2, 4, 8

# struct oval_obj_desc (64 bytes) and struct oval_obj_result (16 bytes)
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
    ]
)
RESULT_DTYPE = np.dtype(
    [
        ("first_bad_fields", "<u4"),
        ("first_bad_unit", "<u4"),
        ("bad_units", "<u4"),
        ("strict_first_bad_unit", "<u4"),
    ]
)
# struct oval_obj_rec (12 bytes) and struct oval_obj_segment (48 bytes): a batch in compact form
REC_DTYPE = np.dtype([("obj_id", "<u4"), ("write_count", "<u4"), ("n_units", "<u4")])
SEGMENT_DTYPE = np.dtype(
    [
        ("bucket", "u1", (16,)),
        ("key_id_base", "<u8"),
        ("unix_micro", "<u8"),
        ("worker_id", "<u4"),
        ("first_obj", "<u4"),
        ("n_obj", "<u4"),
        ("bucket_len", "u1"),
        ("reserved", "u1"),
        ("flags", "<u2"),
    ]
)
assert DESC_DTYPE.itemsize == 64 and RESULT_DTYPE.itemsize == 16
assert REC_DTYPE.itemsize == 12 and SEGMENT_DTYPE.itemsize == 48


class OvalError(RuntimeError):
    """A failure of the engine itself (an oval_status < 0), not a Go `error`."""


_lib = None

# name -> (restype, argtypes); every function declared in include/*.h
_vp, _u64, _i64, _sz, _cp, _int = C.c_void_p, C.c_uint64, C.c_int64, C.c_size_t, C.c_char_p, C.c_int
_pp = C.POINTER(C.c_void_p)
SIGNATURES = {
    # oval_hip.h
    "oval_last_error": (_cp, []),
    "oval_version": (_cp, []),
    "oval_device_count": (_int, [C.POINTER(_int)]),
    "oval_set_device_timeout_ms": (_int, [_u64]),
    "oval_set_small_object_bytes": (_int, [_u64, _u64]),
    "oval_set_big_object_bytes": (_int, [_u64]),
    "oval_ctx_create": (_int, [_int, _vp, _pp]),
    "oval_ctx_destroy": (_int, [_vp]),
    "oval_ctx_sync": (_int, [_vp]),
    "oval_ctx_stream": (_int, [_vp, _pp]),
    "oval_ctx_set_tuning": (_int, [_vp, _int, _int]),
    "oval_dev_alloc": (_int, [_vp, _sz, _pp]),
    "oval_dev_free": (_int, [_vp, _vp]),
    "oval_pinned_alloc": (_int, [_vp, _sz, _pp]),
    "oval_pinned_free": (_int, [_vp, _vp]),
    "oval_copy_h2d": (_int, [_vp, _vp, _vp, _sz]),
    "oval_copy_d2h": (_int, [_vp, _vp, _vp, _sz]),
    "oval_desc_init": (_int, [_vp, _cp, _sz, _cp, _sz, _i64, _i64, _u64, _u64, C.c_uint32, C.c_uint32]),
    "oval_layout_create": (_int, [_vp, _vp, _u64, _pp]),
    "oval_layout_reset": (_int, [_vp, _vp, _vp, _u64]),
    "oval_layout_reset_ahead": (_int, [_vp, _vp, _vp, _u64]),
    "oval_layout_reset_compact": (_int, [_vp, _vp, _vp, _u64, _vp, _u64, C.c_uint32]),
    "oval_layout_read_descs": (_int, [_vp, _vp, _u64, _u64, _vp]),
    "oval_segment_init": (_int, [_vp, _cp, _sz, _i64, _u64, _u64, C.c_uint32, C.c_uint32]),
    "oval_layout_destroy": (_int, [_vp, _vp]),
    "oval_layout_info": (_int, [_vp, C.POINTER(_u64), C.POINTER(_u64), C.POINTER(_u64)]),
    "oval_layout_launches": (_int, [_vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "oval_fill": (_int, [_vp, _vp, _vp, _u64]),
    "oval_validate": (_int, [_vp, _vp, _vp, _u64, _vp, _vp]),
    "oval_validate_noalloc": (_int, [_vp, _vp, _u64, _u64, _vp, _u64, _vp, _vp]),
    "oval_read_unit": (_int, [_vp, _vp, _u64, _vp]),
    "oval_fill_batch_host": (_int, [_vp, _vp, _u64, _vp, _u64]),
    "oval_validate_batch_host": (_int, [_vp, _vp, _u64, _vp, _u64, _vp, _vp]),
    "oval_comm_unique_id": (_int, [_vp]),
    "oval_comm_create": (_int, [_vp, _int, _int, _vp, _pp]),
    "oval_comm_destroy": (_int, [_vp]),
    "oval_comm_info": (_int, [_vp, C.POINTER(_int), C.POINTER(_int)]),
    "oval_allreduce_counters": (_int, [_vp, _vp, _vp, _sz]),
    "oval_gather_first_bad": (_int, [_vp, _vp, _u64, _vp]),
    "oval_allreduce_first_bad": (_int, [_vp, _vp, _vp, _sz]),
    "oval_timer_create": (_int, [_vp, _pp]),
    "oval_timer_destroy": (_int, [_vp]),
    "oval_timer_start": (_int, [_vp, _vp]),
    "oval_timer_stop": (_int, [_vp, _vp]),
    "oval_timer_elapsed_ms": (_int, [_vp, C.POINTER(C.c_float)]),
    # oval_pattern.h
    "oval_pattern_init": (_int, [_int]),
    "oval_pattern_init_engines": (_int, [C.POINTER(_int), _int]),
    "oval_pattern_engine_count": (_int, []),
    "oval_pattern_set_device_timeout": (None, [_u64]),
    "oval_pattern_set_caller_spin": (None, [_int, _int]),
    "oval_pattern_test_lose_flags": (None, [_int]),
    "oval_pattern_quarantined_slots": (_int, []),
    "oval_pattern_shutdown": (None, []),
    "oval_pattern_set_clock": (None, [_i64]),
    "oval_pattern_seed": (None, [_u64]),
    "oval_pattern_decide_size": (_int, [_i64, _i64, C.POINTER(_i64), _cp, _sz]),
    "oval_pattern_decide_size_y": (_int, [_i64, _i64, C.c_double, C.POINTER(_i64), _cp, _sz]),
    "oval_pattern_generate": (_int, [_i64, _i64, _cp, _sz, _cp, _sz, _i64, _vp, _cp, _sz]),
    "oval_pattern_generate_data_unit": (_int, [_i64, _i64, _cp, _sz, _cp, _sz, _i64, _vp, _cp, _sz]),
    "oval_pattern_valid": (_int, [_i64, _cp, _sz, _cp, _sz, _i64, _i64, _vp, _sz, _cp, _sz]),
    "oval_pattern_valid_data_unit": (_int, [_i64, _i64, _cp, _sz, _cp, _sz, _i64, _vp, _cp, _sz]),
    "oval_pattern_generate_batch": (_int, [_sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz]),
    "oval_pattern_valid_batch": (_int, [_sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz]),
    "oval_pattern_valid_stream": (_int, [_i64, _cp, _sz, _cp, _sz, _i64, _i64, _vp, _vp, _cp, _sz]),
    "oval_pattern_set_stream_chunk": (None, [_sz]),
    "oval_pattern_generate_parts": (_int, [_i64, _i64, _cp, _sz, _cp, _sz, _i64, _i64, _pp, _cp, _sz]),
    "oval_parts_count": (_i64, [_vp]),
    "oval_parts_get": (_int, [_vp, _i64, _pp, C.POINTER(_sz), _cp, _sz]),
    "oval_parts_release": (_int, [_vp, _i64]),
    "oval_parts_close": (None, [_vp]),
    "oval_parts_pool_stats": (None, [C.POINTER(_sz), C.POINTER(_sz)]),
    "oval_pattern_generate_begin": (_int, [_i64, _i64, _cp, _sz, _cp, _sz, _i64, _vp, _pp]),
    "oval_pattern_valid_begin": (_int, [_i64, _cp, _sz, _cp, _sz, _i64, _i64, _vp, _sz, _pp]),
    "oval_pattern_poll": (_int, [_vp, _cp, _sz]),
    "oval_pattern_set_batching": (None, [_int, _int, _sz]),
    "oval_pattern_set_lanes": (None, [_int, _int]),
    "oval_pattern_set_wake_tuning": (None, [_int, _int, _int]),
    "oval_pattern_set_direct_callers": (None, [_int]),
    "oval_pattern_batching_stats": (None, [C.POINTER(_u64), C.POINTER(_u64)]),
    "oval_pattern_dump": (_sz, [_cp, _sz, _cp, _sz]),
    "oval_pattern_hex_dump": (_sz, [_vp, _sz, _cp, _sz]),
    "oval_object_generate_key": (_sz, [_i64, _cp, _sz]),
    # oval_context.h
    "oval_execctx_new": (_int, [_cp, _vp, _sz, _i64, _i64, _i64, _i64, _i64, _i64, _pp, _cp, _sz]),
    "oval_execctx_load": (_int, [_cp, _sz, _pp, _cp, _sz]),
    "oval_execctx_load_file": (_int, [_cp, _pp, _cp, _sz]),
    "oval_execctx_marshal": (_sz, [_vp, _vp, _sz]),
    "oval_execctx_save_file": (_int, [_vp, _cp, _cp, _sz]),
    "oval_execctx_destroy": (None, [_vp]),
    "oval_execctx_info": (None, [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64),
                                 C.POINTER(_i64), C.POINTER(_sz), C.POINTER(_sz)]),
    "oval_execctx_endpoint": (_sz, [_vp, _vp, _sz]),
    "oval_execctx_worker_id": (_i64, [_vp, _sz]),
    "oval_execctx_worker_buckets": (_sz, [_vp, _sz]),
    "oval_execctx_bucket_name": (_sz, [_vp, _sz, _sz, _vp, _sz]),
    "oval_execctx_objmeta": (_vp, [_vp, _sz, _sz]),
    "oval_execctx_count_existing": (_u64, [_vp]),
    "oval_objmeta_new": (_vp, [_i64, _i64]),
    "oval_objmeta_free": (None, [_vp]),
    "oval_objmeta_get_random_object": (_i64, [_vp]),
    "oval_objmeta_register_to_existing_list": (_int, [_vp, _cp, _sz, _cp, _sz]),
    "oval_objmeta_pop_existing_random_object": (_i64, [_vp]),
    "oval_objmeta_get_existing_random_object": (_i64, [_vp]),
    "oval_objmeta_exist": (_int, [_vp, _cp, _sz, C.POINTER(_int), _cp, _sz]),
    "oval_objmeta_get_head_and_tail_key": (None, [_vp, _vp, _vp]),
    "oval_objmeta_tidy_up": (_int, [_vp, _cp, _sz]),
    "oval_objmeta_num_objects": (_sz, [_vp]),
    "oval_objmeta_num_existing": (_sz, [_vp]),
    "oval_objmeta_existing_id": (_i64, [_vp, _sz]),
    "oval_objmeta_key_id_offset": (_i64, [_vp]),
    "oval_objmeta_key_prefix": (_sz, [_vp, _vp, _sz]),
    "oval_objmeta_get_object": (_int, [_vp, _i64, _vp, C.POINTER(_i64), C.POINTER(_i64)]),
    "oval_objmeta_set_object": (_int, [_vp, _i64, _i64, _i64]),
    "oval_object_seed_random": (None, [_u64]),
    "oval_execctx_populate": (_int, [_vp, _u64, C.c_double, _i64, _cp, _sz]),
    "oval_execctx_build_descs": (_int, [_vp, _u64, _sz, _sz, _vp, _vp, _u64, C.POINTER(_u64), C.POINTER(_u64),
                                        _cp, _sz]),
    "oval_execctx_build_compact": (_int, [_vp, _u64, _sz, _sz, _vp, _u64, _vp, _u64, C.POINTER(_u64), C.POINTER(_u64),
                                          C.POINTER(_u64), _cp, _sz]),
}


def lib():
    """The loaded library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OvalError(
                f"{LIB_PATH} is missing: build it with `make -C oval_amd/csrc` "
                "(python -c 'import __graft_entry__ as g; g.build()'). "
                "oval_amd has no fallback implementation."
            )
        # A process must end up with ONE HIP runtime.  torch carries its own copy; loaded after this library
        # (which resolves the system's), it initialises a second one and this library then finds no device.
        # Loaded first, torch's copy serves both.  So torch goes first wherever there is a torch.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def last_error() -> str:
    return lib().oval_last_error().decode("utf-8", "replace")


def check(rc: int) -> None:
    if rc != 0:
        raise OvalError(f"oval status {rc}: {last_error()}")

"""CPU-only checks of the product's host side: the C-ABI library loads and
exports everything include/*.h declares, descriptor formatting follows the
reference's rules, and the host-only functions agree with the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from oval_amd import _lib, pattern
from oval_amd import object as obj
from oval_amd.batch import build_descs, make_keys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T0 = 1709558129112997


def _declared_functions():
    names = []
    for h in sorted(os.listdir(os.path.join(ROOT, "include"))):
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names += re.findall(r"\b(oval_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    declared = _declared_functions()
    assert len(declared) >= 40
    for name in declared:
        assert hasattr(L, name), f"{name} is declared in include/ but not exported"
    assert set(declared) == set(_lib.SIGNATURES), "ctypes table and headers disagree"


def test_struct_layouts():
    assert _lib.DESC_DTYPE.itemsize == 64
    assert _lib.DESC_DTYPE.fields["write_count"][1] == 28
    assert _lib.DESC_DTYPE.fields["first_unit"][1] == 32
    assert _lib.DESC_DTYPE.fields["unix_micro"][1] == 40
    assert _lib.DESC_DTYPE.fields["body_offset"][1] == 48
    assert _lib.RESULT_DTYPE.itemsize == 16


def _desc_init(bucket, key, wc, wid, t=T0, off=0, first=0, n=1):
    d = np.zeros(1, dtype=_lib.DESC_DTYPE)
    b = bucket if isinstance(bucket, bytes) else bucket.encode()
    k = key if isinstance(key, bytes) else key.encode()
    rc = _lib.lib().oval_desc_init(d.ctypes.data, b, len(b), k, len(k), wc, wid, t, off, first, n)
    return rc, d[0]


def test_desc_init_is_the_unit_header(oracle):
    rc, d = _desc_init("test-bucket", "ov00020000e2", 1, 0x561B, first=0)
    assert rc == 0
    unit0 = oracle.generate_data_unit(0, 0x561B, "test-bucket", "ov00020000e2", 1, T0)
    assert d.tobytes()[:48] == unit0[:48]
    assert (d["bucket_len"], d["key_len"], d["flags"]) == (11, 12, 0)


def test_desc_init_rules(oracle):
    # bucket is cut to 16 bytes (pattern.go:74-76), key never is
    rc, d = _desc_init("test-long-long-long-bucket", "test-key", 300, 100)
    assert rc == 0 and bytes(d["bucket"]) == b"test-long-long-l" and d["bucket_len"] == 16
    assert bytes(d["key"]) == b"test-key    " and d["key_len"] == 8
    rc, _ = _desc_init("b", "0123456789abc", 1, 1)
    assert rc == -5 and "12" in _lib.last_error()
    # uint32 truncation of write count / worker (pattern.go:101,105)
    rc, d = _desc_init("b", "k", (1 << 32) + 5, 7)
    assert rc == 0 and d["write_count"] == 5
    rc, d = _desc_init("b", "k", 1, (1 << 32) + 9)
    assert d["worker_id"] == 9 and d["flags"] & _lib.DESC_WORKER_OOR
    rc, d = _desc_init("b", "k", 1, -1)
    assert d["flags"] & _lib.DESC_WORKER_OOR
    # names with white space at an edge can never equal their trimmed field
    for name, flag in ((" b", 1), ("b\t", 1), ("b ", 1), ("a b", 0), ("", 0)):
        rc, d = _desc_init(name, "k", 1, 1)
        assert rc == 0 and (d["flags"] & _lib.DESC_BUCKET_SLOW) == flag, name
    rc, d = _desc_init("b", "k ", 1, 1)
    assert d["flags"] & _lib.DESC_KEY_SLOW
    # multi-byte runes: fmt pads by rune count, the unit would be 257 bytes
    rc, d = _desc_init("café", "k", 1, 1)
    assert rc == 0 and d["flags"] & _lib.DESC_NO_FILL
    assert len(oracle.generate_data_unit(0, 1, "café", "k", 1, T0)) == 257
    rc, _ = _desc_init("b", "k", 1, 1, off=100)
    assert rc == -1


def test_build_descs_matches_desc_init():
    ids = np.array([0, 1, (2 << 24) + 0xE2, (1 << 32) + (3 << 24) + 0xFF], dtype=np.uint64)
    descs = build_descs("test-bucket", make_keys(ids), [1, 2, 3, 4], [10, 11, 12, 13], T0, [512, 256, 1024, 0])
    off = 0
    for i, oid in enumerate(ids):
        rc, d = _desc_init("test-bucket", obj.generateKey(int(oid)), i + 1, 10 + i, off=off, n=[2, 1, 4, 0][i])
        assert rc == 0 and d.tobytes() == descs[i].tobytes()
        off += [512, 256, 1024, 0][i]


def test_generate_key(oracle):
    for oid in (0, 1, 0xFFFFFF, (2 << 24) + 0xE2, (255 << 32) + (254 << 24) + 0xABCDEF, (1 << 40) - 1):
        assert obj.generateKey(oid) == oracle.generate_key(oid)
        assert len(obj.generateKey(oid)) == 12
    assert obj.NewObject(5) == obj.Object("ov0000000005", 0, 0)
    o = obj.Object("k", 10, 3); o.Clear()
    assert (o.Size, o.WriteCount) == (0, 0)
    assert (obj.MaxBucketNameLength, obj.MaxKeyLength, obj.KeyShortPrefix, obj.KeyPrefixLength) == (16, 12, "ov", 6)


def test_decide_size_agrees_with_oracle(oracle, golden):
    rng = np.random.default_rng(11)
    ys = list(rng.random(3000)) + [0.0, 1 - 2.0 ** -52, 1 - 2.0 ** -53, 0.5]
    for lo, hi in ((4096, 16384), (4096, 1 << 20), (256, 1024), (512, 512), (1024, 5 << 20)):
        for y in ys:
            assert pattern.DecideSize(lo, hi, float(y)) == oracle.decide_size(lo, hi, float(y))
    for tc in golden["pattern_test"]["decide_size"]:            # pattern_test.go:202-255
        if tc["err"]:
            with pytest.raises(pattern.PatternError) as e:
                pattern.DecideSize(tc["min"], tc["max"])
            with pytest.raises(oracle.OracleError) as e2:
                oracle.decide_size(tc["min"], tc["max"], 0.1)
            assert str(e.value) == str(e2.value)
        else:
            pattern.seed(3)
            for _ in range(200):
                s = pattern.DecideSize(tc["min"], tc["max"])
                assert tc["min"] <= s <= tc["max"] and s % 256 == 0


def test_decide_size_seeded_stream_is_reproducible():
    pattern.seed(42)
    a = [pattern.DecideSize(4096, 16384) for _ in range(50)]
    pattern.seed(42)
    assert a == [pattern.DecideSize(4096, 16384) for _ in range(50)]
    assert len(set(a)) > 10


def test_dump_and_hexdump_agree_with_oracle(oracle):
    rng = np.random.default_rng(2)
    for n in (0, 1, 7, 8, 9, 15, 16, 17, 31, 32, 40, 47, 48, 79, 188, 255, 256):
        data = rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
        hd = pattern.hexDump(data)
        assert hd == oracle.hex_dump(data)
        assert pattern.dump(hd) == oracle.dump(hd.encode("latin-1"))


def test_generate_size_errors_need_no_device(oracle):
    o = obj.Object("0123456789abc", 0, 1)
    for size, key in ((512, "0123456789abc"), (300, "k"), (255, "k"), (1024, "café-key")):
        o.Key = key
        with pytest.raises(pattern.PatternError) as e:
            pattern.Generate(size, 1, "test-bucket", o)
        with pytest.raises(oracle.OracleError) as e2:
            oracle.generate(size, 1, "test-bucket", key, 1, T0)
        assert str(e.value) == str(e2.value)
    o.Key = "k"
    assert pattern.Generate(0, 1, "b", o) == b""
    o.Size = 0
    pattern.Valid(1, "b", o, b"")          # zero units: nil without touching the device


def test_no_silent_cpu_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    o = obj.Object("k", 256, 1)
    with pytest.raises(_lib.OvalError, match="no HIP device|no CPU path"):
        pattern.Generate(256, 1, "b", o)
    with pytest.raises(_lib.OvalError):
        pattern.Valid(1, "b", o, bytes(256))
    h = C.c_void_p()
    assert _lib.lib().oval_ctx_create(0, None, C.byref(h)) == -7


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "oval_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in text.lower(), f"{f} mentions the oracle"


def test_no_cpp_exception_leaves_the_c_abi():
    """A host in another language (cgo) can only abort on a C++ exception: every
    entry point stops them at the boundary and returns a status.  Item counts no
    allocation can satisfy reach the library's vectors before anything else."""
    L = _lib.lib()
    for n, want in ((1 << 60, (_lib.OVAL_E_INTERNAL, _lib.OVAL_E_NOMEM)), (1 << 42, (_lib.OVAL_E_NOMEM, _lib.OVAL_E_INTERNAL))):
        rc = L.oval_pattern_generate_batch(C.c_size_t(n), None, None, None, None, None, None, None, None, None, None,
                                           C.c_size_t(0))
        assert rc in want, rc
        assert "inside the library" in _lib.last_error()
    # an entry point with an error buffer of the caller's gets the text there as well
    err, out = C.create_string_buffer(b"\xff" * 255, 256), np.zeros(4096, dtype=np.uint8)
    rc = L.oval_pattern_generate(4096, 1, b"b", C.c_size_t(1 << 62), b"k", 1, 1, out.ctypes.data, err, 256)   # a name of 2^62 bytes
    assert rc in (_lib.OVAL_E_INTERNAL, _lib.OVAL_E_NOMEM), rc
    assert b"inside the library" in err.value
    # the library is still usable afterwards
    assert pattern.DecideSize(4096, 4096) == 4096

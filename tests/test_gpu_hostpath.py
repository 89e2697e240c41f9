"""Bodies that start and end in host memory: the one-shot host entry points
(pinned and pageable buffers, layouts with gaps), Generate whose result leaves
in parts (the reference's multipartUpload, internal/s3client/s3client.go:164-236)
and Valid over a 256 MiB body behind a reader that dribbles
(internal/pattern/pattern.go:118-134; body from s3client.go:238-251).

Every check is against the CPU oracle, through the C ABI.
"""
import ctypes as C
import io

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

T0 = 1709558129112997


@pytest.fixture(scope="module")
def ctx():
    from oval_amd import engine

    c = engine.Context(0)
    yield c
    c.close()


def _pinned(ctx, nbytes):
    from oval_amd import _lib

    p = C.c_void_p()
    _lib.check(_lib.lib().oval_pinned_alloc(ctx.handle, nbytes, C.byref(p)))
    return p, np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(nbytes,))


def _gappy_descs(n_obj, gap, first_offset, seed):
    from oval_amd.batch import build_descs, decide_sizes, make_keys

    rng = np.random.default_rng(seed)
    sizes = decide_sizes(rng.random(n_obj), 4096, 65536)
    d = build_descs("test-bucket", make_keys(np.arange(n_obj) + (3 << 24)), rng.integers(1, 1000, n_obj),
                    np.full(n_obj, 0x561B), T0, sizes, gap_bytes=gap)
    d["body_offset"] += np.uint64(first_offset)
    return d


@pytest.mark.parametrize("kind", ["pinned", "pageable"])
@pytest.mark.parametrize("gap,first_offset,n_obj", [(512, 768, 3000),        # a few dozen runs per slice
                                                     (256, 0, 6000),
                                                     (0, 4096, 2000),         # one run, not at offset 0
                                                     (1 << 20, 256, 300)])    # few objects, wide gaps
def test_host_fill_leaves_gaps_untouched(ctx, oracle, kind, gap, first_offset, n_obj):
    """Only bytes that belong to an object may change in the caller's buffer:
    gaps between bodies and the space before the first one keep their poison,
    whatever earlier calls left in the staging ring."""
    from oval_amd import _lib
    from oval_amd.batch import apply_corruption_numpy, plan_corruption, span_bytes

    descs = _gappy_descs(n_obj, gap, first_offset, seed=n_obj + gap)
    n = span_bytes(descs) + 1000
    if kind == "pinned":
        p, host = _pinned(ctx, n)
    else:
        host = np.empty(n, dtype=np.uint8)
    try:
        # leave something else in the ring first
        other = _gappy_descs(500, 0, 0, seed=1)
        scratch = np.empty(span_bytes(other), dtype=np.uint8)
        ctx.fill_batch_host(other, scratch)
        host[:] = 0xA5
        ctx.fill_batch_host(descs, host)
        exp = np.full(n, 0xA5, dtype=np.uint8)
        oracle.fill_batch(descs.view(oracle.DESC_DTYPE), exp)
        bad = np.flatnonzero(host != exp)
        assert bad.size == 0, f"first differing byte {bad[:4]} of {n}"
        apply_corruption_numpy(host, descs, plan_corruption(descs, 2e-4, seed=4))
        res, cnt = ctx.validate_batch_host(descs, host)
        eres, ecnt = oracle.validate_batch(descs.view(oracle.DESC_DTYPE), host)
        for f in eres.dtype.names:
            assert np.array_equal(res[f], eres[f]), f
        assert np.array_equal(cnt, ecnt) and ecnt[1] > 0
    finally:
        if kind == "pinned":
            _lib.check(_lib.lib().oval_pinned_free(ctx.handle, p))


def test_host_calls_reuse_the_contexts_tables(ctx, oracle):
    """Batches of different sizes one after the other through one context:
    larger, smaller, empty, larger again (tables grow, never shrink)."""
    from oval_amd.batch import span_bytes

    for n_obj, seed in ((100, 1), (5000, 2), (3, 3), (0, 4), (9000, 5), (1, 6)):
        descs = _gappy_descs(n_obj, 0, 0, seed)
        n = max(span_bytes(descs), 256)
        host = np.full(n, 0x11, dtype=np.uint8)
        ctx.fill_batch_host(descs, host)
        exp = np.full(n, 0x11, dtype=np.uint8)
        oracle.fill_batch(descs.view(oracle.DESC_DTYPE), exp)
        assert np.array_equal(host, exp)
        if n_obj:
            host[int(descs["body_offset"][n_obj // 2]) + 29] ^= 0x40
        res, cnt = ctx.validate_batch_host(descs, host)
        eres, ecnt = oracle.validate_batch(descs.view(oracle.DESC_DTYPE), host)
        for f in eres.dtype.names:
            assert np.array_equal(res[f], eres[f]), (n_obj, f)
        assert np.array_equal(cnt, ecnt)


def test_the_streaming_kernels_run_straight_on_pinned_host_memory(ctx, oracle):
    """oval_fill / oval_validate take any pointer the GPU can address: handed a
    pinned host buffer as the body, the kernels' own stores and loads cross the
    link, with no staging in between.  Bytes, gaps, verdicts and counters are the
    oracle's."""
    import torch

    from oval_amd import _lib, engine
    from oval_amd.batch import apply_corruption_numpy, plan_corruption, span_bytes

    descs = _gappy_descs(4000, 512, 1024, seed=77)
    n = span_bytes(descs) + 512
    p, host = _pinned(ctx, n)
    try:
        host[:] = 0x3C
        layout = engine.Layout(ctx, descs)
        _lib.check(_lib.lib().oval_fill(ctx.handle, layout._h, p, n))
        ctx.sync()
        exp = np.full(n, 0x3C, dtype=np.uint8)
        oracle.fill_batch(descs.view(oracle.DESC_DTYPE), exp)
        assert np.array_equal(host, exp)
        apply_corruption_numpy(host, descs, plan_corruption(descs, 5e-4, seed=9))
        results = torch.empty((len(descs), 4), dtype=torch.int32, device="cuda:0")
        counters = torch.zeros(engine.NUM_COUNTERS, dtype=torch.int64, device="cuda:0")
        torch.cuda.synchronize()
        _lib.check(_lib.lib().oval_validate(ctx.handle, layout._h, p, n, results.data_ptr(), counters.data_ptr()))
        ctx.sync()
        res = engine.results_to_numpy(results, len(descs))
        eres, ecnt = oracle.validate_batch(descs.view(oracle.DESC_DTYPE), host)
        for f in eres.dtype.names:
            assert np.array_equal(res[f], eres[f]), f
        assert np.array_equal(counters.cpu().numpy()[:len(ecnt)], ecnt) and ecnt[1] > 0
        layout.close()
    finally:
        _lib.check(_lib.lib().oval_pinned_free(ctx.handle, p))


# ---------------------------------------------------------------------------
# Generate in parts
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("size,part", [(8192, 4096), (8192, 3000), (256, 1 << 20), (200 * 1024, 65536),
                                       (96 * (1 << 20) + 256 * 7, 5 * (1 << 20)),     # --multipart_thresh 5m
                                       (96 * (1 << 20) + 256 * 7, 33 * (1 << 20) + 1),
                                       (40 * (1 << 20), 100 * (1 << 20))])
def test_generate_parts_concatenate_to_generate(oracle, size, part):
    from oval_amd import pattern
    from oval_amd.object import Object

    pattern.init(0)
    pattern.set_clock(T0)
    try:
        b, wid = "test-long-long-long-bucket", 0x561B            # cut to 16 bytes like pattern.go:74-76
        obj = Object("ov00020000e2", size, 7)
        want = np.frombuffer(oracle.generate(size, wid, b, obj.Key, obj.WriteCount, T0), dtype=np.uint8)
        with pattern.GenerateParts(size, wid, b, obj, part) as parts:
            n = len(parts)
            assert n == -(-size // part)                          # the UploadPart calls of s3client.go:176-211
            order = list(range(n))
            order = order[1::2] + order[0::2]                     # parts may be fetched and released in any order
            for i in order:
                view = parts.part(i)
                lo, hi = i * part, min((i + 1) * part, size)
                assert view.size == hi - lo
                assert np.array_equal(view, want[lo:hi]), f"part {i}"
                parts.release(i)
            with pytest.raises(Exception):
                parts.part(0)                                     # released
    finally:
        pattern.set_clock(-1)


def test_generate_parts_errors_and_pool(oracle):
    from oval_amd import pattern
    from oval_amd._lib import OvalError
    from oval_amd.object import Object

    pattern.init(0)
    pattern.set_clock(T0)
    try:
        obj = Object("ov0000000001", 0, 1)
        with pytest.raises(pattern.PatternError) as e:            # pattern.go:87-89
            pattern.GenerateParts(1000, 1, "b", obj, 512)
        with pytest.raises(oracle.OracleError) as oe:
            oracle.generate(1000, 1, "b", obj.Key, 1, T0)
        assert str(e.value) == str(oe.value)
        with pytest.raises(pattern.PatternError) as e:
            pattern.GenerateParts(512, 1, "b", Object("a-key-of-17-bytes", 0, 1), 512)
        with pytest.raises(oracle.OracleError) as oe:
            oracle.generate(512, 1, "b", "a-key-of-17-bytes", 1, T0)
        assert str(e.value) == str(oe.value)
        with pytest.raises(OvalError):
            pattern.GenerateParts(512, 1, "b", obj, 0)
        empty = pattern.GenerateParts(0, 1, "b", obj, 512)
        assert len(empty) == 0
        empty.close()
        # the abort path: close with parts outstanding, then the buffer is reused
        size = 48 << 20
        p1 = pattern.GenerateParts(size, 1, "b", obj, 8 << 20)
        first = p1.part(0).copy()
        p1.close()
        idle, idle_bytes = pattern.parts_pool_stats()
        assert idle >= 1 and idle_bytes >= size
        p2 = pattern.GenerateParts(size - 4096, 2, "b", obj, 8 << 20)
        assert pattern.parts_pool_stats()[0] == idle - 1          # took the idle buffer instead of pinning a new one
        got = np.concatenate([p2.part(i) for i in range(len(p2))])
        p2.close()
        assert np.array_equal(got, np.frombuffer(oracle.generate(size - 4096, 2, "b", obj.Key, 1, T0), np.uint8))
        assert np.array_equal(first, np.frombuffer(oracle.generate(8 << 20, 1, "b", obj.Key, 1, T0), np.uint8))
    finally:
        pattern.set_clock(-1)


def test_generate_parts_from_many_threads(oracle):
    """Concurrent uploads: each thread generates its own body in parts while the
    others do the same (more uploads than staging slots in flight)."""
    import threading

    from oval_amd import pattern
    from oval_amd.object import Object

    pattern.init(0)
    pattern.set_clock(T0)
    errors = []

    def work(t):
        try:
            size = (3 + t % 5) * (8 << 20) + 256 * t
            obj = Object("ov%010x" % t, size, t + 1)
            want = np.frombuffer(oracle.generate(size, t, "test-bucket", obj.Key, obj.WriteCount, T0), np.uint8)
            with pattern.GenerateParts(size, t, "test-bucket", obj, 5 << 20) as parts:
                for i in range(len(parts)):
                    if not np.array_equal(parts.part(i), want[i * (5 << 20):(i + 1) * (5 << 20)]):
                        errors.append((t, i))
                    parts.release(i)
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    try:
        ts = [threading.Thread(target=work, args=(t,)) for t in range(24)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert not errors, errors[:3]
    finally:
        pattern.set_clock(-1)


# ---------------------------------------------------------------------------
# Valid over a large streamed body
# ---------------------------------------------------------------------------
class _Dribble(io.RawIOBase):
    """A reader over a numpy body that never returns more than `step` bytes."""

    def __init__(self, data, step):
        self.data, self.pos, self.step = data, 0, step

    def read(self, n=-1):
        n = min(n, self.step)
        out = self.data[self.pos:self.pos + n]
        self.pos += len(out)
        return out.tobytes()


def test_valid_streams_a_256_mib_body_behind_a_dribbling_reader(oracle):
    """cfg3-sized object (256 MiB = 1,048,576 units, 8 chunks of 32 MiB) read in
    pieces that straddle units and chunks: the oracle's verdict and text for a
    clean body, for damage in the first, a middle and the last chunk, for two
    damaged chunks (the earlier wins), and for a body that ends early."""
    from oval_amd import pattern
    from oval_amd.object import Object

    pattern.init(0)
    pattern.set_clock(T0)
    try:
        size, wid, b = 256 << 20, 0x561B, "test-bucket"
        obj = Object("ov00020000e2", size, 2)
        with pattern.GenerateParts(size, wid, b, obj, size) as parts:
            body = parts.part(0).copy()
        assert oracle.valid(wid, b, obj.Key, size, obj.WriteCount, body)[0] == 0

        def check(data, step):
            try:
                pattern.ValidStream(wid, b, obj, _Dribble(data, step))
                got = None
            except pattern.PatternError as e:
                got = str(e)
            want = oracle.valid(wid, b, obj.Key, size, obj.WriteCount, data)[2]
            assert got == want
            return got

        assert check(body, 1_000_003) is None                                  # ~270 reads, none aligned
        chunk = 32 << 20
        for unit in (5, chunk // 256 - 1, 3 * (chunk // 256) + 12345, (size // 256) - 1):
            off = unit * 256 + 28                                              # the write count (pattern.go:152-157)
            body[off] ^= 0x01
            text = check(body, 4 << 20)
            assert text is not None and "WriteCount is wrong" in text
            body[off] ^= 0x01
        first, later = 2 * (chunk // 256) + 7, 6 * (chunk // 256) + 9
        body[first * 256 + 33] ^= 0x80                                         # offset field
        body[later * 256 + 3] ^= 0x80                                          # bucket field
        text = check(body, 4 << 20)
        assert "OffsetInObject is wrong" in text and "Bucket name" not in text
        body[first * 256 + 33] ^= 0x80
        body[later * 256 + 3] ^= 0x80
        text = check(body[:5 * chunk + 1000], 4 << 20)                         # short read in chunk 5
        assert text.startswith("could not read some data. (expected: 256byte, actual: 232byte)")
    finally:
        pattern.set_clock(-1)


def test_every_host_path_at_once_while_batching_toggles(oracle):
    """Worker threads mixing small Generate / Valid calls (micro-batched), bodies
    too large to share a slot, multipart generation and streamed validation, while
    another thread switches micro-batching on and off: every call returns what
    the oracle says, whichever path it happened to take."""
    import threading
    import time

    from oval_amd import pattern
    from oval_amd.object import Object

    pattern.init(0)
    pattern.set_clock(T0)
    pattern.set_stream_chunk(1 << 20)
    stop = threading.Event()
    errors = []

    def toggler():
        on = False
        while not stop.is_set():
            on = not on
            pattern.set_batching(on, window_us=0)
            time.sleep(0.003)
        pattern.set_batching(False)

    def work(t):
        rng = np.random.default_rng(1000 + t)
        try:
            for it in range(40):
                kind = (t + it) % 4
                wc, wid, b = int(rng.integers(1, 500)), 0x561B + t, "test-bucket"
                key = "ov%010x" % (t * 1000 + it)
                if kind == 0:                                           # small, shares launches
                    size = 256 * int(rng.integers(1, 64))
                elif kind == 1:                                         # too large for a shared slot
                    size = 256 * int(rng.integers(1100, 3000))
                else:
                    size = 256 * int(rng.integers(5000, 20000))         # 1.2 - 5 MiB
                obj = Object(key, size, wc)
                want = oracle.generate(size, wid, b, key, wc, T0)
                if kind == 2:
                    with pattern.GenerateParts(size, wid, b, obj, 1 << 20) as parts:
                        body = b"".join(parts.part(i).tobytes() for i in range(len(parts)))
                else:
                    body = pattern.Generate(size, wid, b, obj)
                if body != want:
                    errors.append((t, it, "generate"))
                    continue
                data = bytearray(body)
                if it % 3 == 0:
                    data[int(rng.integers(0, size))] ^= 1 << int(rng.integers(0, 8))
                if it % 5 == 0:
                    data = data[:int(rng.integers(0, size))]
                data = bytes(data)
                try:
                    if kind == 3:
                        pattern.ValidStream(wid, b, obj, io.BytesIO(data))    # chunks of 1 MiB through the double buffer
                    else:
                        pattern.Valid(wid, b, obj, data)
                    got = None
                except pattern.PatternError as e:
                    got = str(e)
                if got != oracle.valid(wid, b, key, size, wc, data)[2]:
                    errors.append((t, it, "valid"))
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    tg = threading.Thread(target=toggler)
    ts = [threading.Thread(target=work, args=(t,)) for t in range(12)]
    try:
        tg.start()
        for t in ts:
            t.start()
        for t in ts:
            t.join()
    finally:
        stop.set()
        tg.join()
        pattern.set_stream_chunk(32 << 20)
        pattern.set_clock(-1)
    assert not errors, errors[:5]


def test_idle_generated_bodies_do_not_starve_other_callers(oracle):
    """More multipart bodies open at once than the engine has staging slots, none
    of them touched yet (uploads that have not started): further bodies and
    micro-batched small calls still go through, because a body's slot is taken
    back as soon as its last slice has landed, whoever notices."""
    from oval_amd import pattern
    from oval_amd.object import Object

    pattern.init(0)
    pattern.set_clock(T0)
    held = []
    try:
        size = 3 << 20
        for i in range(48):                                               # the engine has 34 slots
            obj = Object("ov%010x" % i, size, i + 1)
            held.append((obj, pattern.GenerateParts(size, 7, "test-bucket", obj, 1 << 20)))
        pattern.set_batching(True, window_us=0)
        small = Object("ov00000000ff", 4096, 3)
        for direct in (0, 16):                                             # through the micro-batcher, then launching on its own
            pattern.set_direct_callers(direct)
            for _ in range(50):
                assert pattern.Generate(4096, 7, "test-bucket", small) == oracle.generate(4096, 7, "test-bucket", small.Key, 3, T0)
        pattern.set_batching(False)
        big = Object("ov00000000fe", 1 << 20, 2)                          # takes a slot of its own
        pattern.Valid(7, "test-bucket", big, pattern.Generate(1 << 20, 7, "test-bucket", big))
        for obj, parts in held[::7]:
            want = np.frombuffer(oracle.generate(size, 7, "test-bucket", obj.Key, obj.WriteCount, T0), np.uint8)
            got = np.concatenate([parts.part(i) for i in range(len(parts))])
            assert np.array_equal(got, want)
    finally:
        pattern.set_direct_callers(16)
        pattern.set_batching(False)
        for _, parts in held:
            parts.close()
        pattern.set_clock(-1)


def test_shutdown_with_a_generated_body_still_open(oracle):
    """Shutdown does not wait for a caller to finish reading a multipart body
    whose bytes have all landed; the body stays readable (it lives in pinned host
    memory), and the engine comes back with the next init."""
    from oval_amd import pattern
    from oval_amd.object import Object

    pattern.init(0)
    pattern.set_clock(T0)
    try:
        size = 5 << 20
        obj = Object("ov00000000aa", size, 4)
        parts = pattern.GenerateParts(size, 9, "test-bucket", obj, 1 << 20)
        first = parts.part(0).copy()
        pattern.shutdown()
        want = np.frombuffer(oracle.generate(size, 9, "test-bucket", obj.Key, 4, T0), np.uint8)
        assert np.array_equal(first, want[:1 << 20])
        assert np.array_equal(parts.part(4), want[4 << 20:])
        parts.close()
        pattern.init(0)
        assert pattern.Generate(512, 9, "test-bucket", obj) == oracle.generate(512, 9, "test-bucket", obj.Key, 4, T0)
    finally:
        pattern.init(0)
        pattern.set_clock(-1)


def test_process_exits_cleanly_with_micro_batching_still_on():
    """A caller that never switches micro-batching off (a Go binary that just
    returns from main) must not be aborted at exit by the lane threads it leaves
    asleep."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("from oval_amd import pattern\n"
            "from oval_amd.object import Object\n"
            "pattern.init(0)\n"
            "pattern.set_batching(True)\n"
            "b = pattern.Generate(4096, 1, 'b', Object('ov0000000001', 4096, 1))\n"
            "print('generated', len(b))\n")
    p = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=110)
    assert p.returncode == 0, (p.returncode, p.stderr[-1500:])
    assert "generated 4096" in p.stdout


def test_one_large_object_in_the_callers_memory(oracle):
    """Generate of a body of many megabytes into the caller's own buffer and Valid
    over bytes in memory go through the sliced host-memory path (no staging slot
    of the body's size): the bytes and every verdict are the oracle's — clean, a
    damaged first / middle / last unit, two damaged units, a body cut short inside
    a unit and one cut at a unit's end."""
    from oval_amd import pattern
    from oval_amd.object import Object

    pattern.init(0)
    pattern.set_clock(T0)
    try:
        size = (40 << 20) + 768
        obj = Object("ov00000000c1", size, 5)
        body = pattern.Generate(size, 0x1234, "test-bucket", obj)
        want = oracle.generate(size, 0x1234, "test-bucket", obj.Key, 5, T0)
        assert body == want
        pattern.Valid(0x1234, "test-bucket", obj, body)
        n_units = size // 256
        for units in ([0], [n_units // 2], [n_units - 1], [70000, 9]):
            bad = bytearray(body)
            for u in units:
                bad[u * 256 + 17] ^= 0x20                                   # key field
            bad = bytes(bad)
            with pytest.raises(pattern.PatternError) as e:
                pattern.Valid(0x1234, "test-bucket", obj, bad)
            assert str(e.value) == oracle.valid(0x1234, "test-bucket", obj.Key, size, 5, bad)[2]
        for cut in (size - 100, size - 256, (33 << 20) + 5):
            short = body[:cut]
            with pytest.raises(pattern.PatternError) as e:
                pattern.Valid(0x1234, "test-bucket", obj, short)
            assert str(e.value) == oracle.valid(0x1234, "test-bucket", obj.Key, size, 5, short)[2]
        # damage wins over the short read that follows it (pattern.go:124-131)
        both = bytearray(body[:size - 300]); both[5 * 256 + 28] ^= 1
        with pytest.raises(pattern.PatternError) as e:
            pattern.Valid(0x1234, "test-bucket", obj, bytes(both))
        assert str(e.value) == oracle.valid(0x1234, "test-bucket", obj.Key, size, 5, bytes(both))[2]
        assert "WriteCount is wrong" in str(e.value)
    finally:
        pattern.set_clock(-1)

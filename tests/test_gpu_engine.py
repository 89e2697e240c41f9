"""The device pipeline behind the single-object calls (one engine per GPU,
staging slots, micro-batcher): slot reuse across the batch forms and the shared
launches, waits on the device that end, and two engines in one process.  All
results are held to the CPU oracle."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

T0 = 1709558129112997


def _objects(rng, n, oracle, max_units=64, workers=0x10000):
    from oval_amd.object import Object

    items = []
    for i in range(n):
        key = oracle.generate_key(int(rng.integers(0, 1 << 40)))
        size = int(rng.integers(1, max_units + 1)) * 256
        items.append((size, int(rng.integers(0, workers)), "test-bucket" if i % 3 else "test-long-long-long-bucket",
                      Object(key, size, int(rng.integers(1, 1000)))))
    return items


def _cycle(pattern, it):
    """Generate, Valid (nil), Valid of a damaged copy (the reference's text)."""
    size, wid, bucket, obj = it
    body = pattern.Generate(size, wid, bucket, obj)
    pattern.Valid(wid, bucket, obj, body)
    bad = bytearray(body)
    bad[28] ^= 0xFF
    try:
        pattern.Valid(wid, bucket, obj, bytes(bad))
        text = None
    except pattern.PatternError as e:
        text = str(e)
    return body, text


def _check(oracle, items, results):
    for (size, wid, bucket, obj), (body, text) in zip(items, results):
        assert body == oracle.generate(size, wid, bucket, obj.Key, obj.WriteCount, T0)
        bad = bytearray(body)
        bad[28] ^= 0xFF
        assert text == oracle.valid(wid, bucket, obj.Key, size, obj.WriteCount, bytes(bad))[2]


def test_slot_grown_by_a_batch_form_later_carries_shared_launches(oracle):
    """A GenerateBatch / ValidBatch of about 1,800 small objects grows a staging
    slot's job arrays past what a shared slot needs without ever having been
    one; the same slot must still work when it becomes the micro-batcher's open
    slot afterwards (more than 16 concurrent callers, batching on)."""
    from oval_amd import pattern

    pattern.set_clock(T0)
    rng = np.random.default_rng(1800)
    items = _objects(rng, 1800, oracle, max_units=24)
    out = pattern.GenerateBatch(items)
    for (size, wid, bucket, obj), got in zip(items, out):
        assert got == oracle.generate(size, wid, bucket, obj.Key, obj.WriteCount, T0)
    got = pattern.ValidBatch([(wid, bucket, obj, body) for (size, wid, bucket, obj), body in zip(items, out)])
    assert all(g is None for g in got)
    b0, j0 = pattern.batching_stats()
    pattern.set_direct_callers(0)
    pattern.set_batching(True, window_us=500)
    try:
        with ThreadPoolExecutor(32) as ex:
            results = list(ex.map(lambda it: _cycle(pattern, it), items[:400]))
    finally:
        pattern.set_direct_callers(16)
        pattern.set_batching(False)
    b1, j1 = pattern.batching_stats()
    assert j1 - j0 == 3 * 400 and b1 - b0 < j1 - j0
    _check(oracle, items[:400], results)


def test_a_batch_that_never_finishes_fails_its_call_and_costs_only_its_slot(oracle):
    """A device that stops answering must not take the calling threads with it:
    the wait ends after the device time limit with an engine failure, the staging
    slot of that batch is never used again, and calls on other slots go on."""
    from oval_amd import pattern
    from oval_amd._lib import OvalError

    pattern.set_clock(T0)
    rng = np.random.default_rng(5)
    items = _objects(rng, 40, oracle)
    _cycle(pattern, items[0])                                   # warm: slots exist
    q0 = pattern.quarantined_slots()
    pattern.set_device_timeout(400)
    try:
        # a call that launches on its own (batching off): its completion flag is never seen
        pattern.test_lose_flags(1)
        size, wid, bucket, obj = items[1]
        with pytest.raises(OvalError, match="did not finish the batch"):
            pattern.Generate(size, wid, bucket, obj)
        assert pattern.quarantined_slots() == q0 + 1
        results = [_cycle(pattern, it) for it in items[2:12]]
        _check(oracle, items[2:12], results)
        # the same under the micro-batcher: every caller of the lost batch fails, nobody hangs, later batches work
        pattern.set_direct_callers(0)
        pattern.set_batching(True, window_us=2000)
        pattern.test_lose_flags(1)

        def attempt(it):
            try:
                return _cycle(pattern, it)
            except OvalError as e:
                return e

        with ThreadPoolExecutor(8) as ex:
            mixed = list(ex.map(attempt, items[12:]))
        failed = [m for m in mixed if isinstance(m, OvalError)]
        assert failed and all("did not finish the batch" in str(f) for f in failed)
        assert pattern.quarantined_slots() == q0 + 2
        good = [(it, m) for it, m in zip(items[12:], mixed) if not isinstance(m, OvalError)]
        _check(oracle, [g[0] for g in good], [g[1] for g in good])
        with ThreadPoolExecutor(8) as ex:
            results = list(ex.map(lambda it: _cycle(pattern, it), items[12:]))
        _check(oracle, items[12:], results)
    finally:
        pattern.test_lose_flags(0)
        pattern.set_device_timeout(30000)
        pattern.set_direct_callers(16)
        pattern.set_batching(False)


def test_two_engines_in_one_process(oracle):
    """One engine per GPU, calls routed by worker ID: on a one-GPU box two engines
    on device 0 stand in for two GPUs.  Single calls (own launch and shared
    launch), the batch forms across both engines, a streamed body and a body
    generated in parts are all held to the oracle."""
    import io

    from oval_amd import pattern
    from oval_amd.object import Object

    pattern.shutdown()
    pattern.init_engines([0, 0])
    try:
        assert pattern.engine_count() == 2
        pattern.set_clock(T0)
        rng = np.random.default_rng(22)
        items = _objects(rng, 300, oracle, workers=64)
        assert {it[1] % 2 for it in items} == {0, 1}                # both engines get work
        _check(oracle, items[:40], [_cycle(pattern, it) for it in items[:40]])
        pattern.set_direct_callers(0)
        pattern.set_batching(True, window_us=500)
        try:
            with ThreadPoolExecutor(32) as ex:
                results = list(ex.map(lambda it: _cycle(pattern, it), items))
        finally:
            pattern.set_direct_callers(16)
            pattern.set_batching(False)
        _check(oracle, items, results)
        # batch forms: one list per engine, run side by side
        out = pattern.GenerateBatch(items)
        for (size, wid, bucket, obj), got in zip(items, out):
            assert got == oracle.generate(size, wid, bucket, obj.Key, obj.WriteCount, T0)
        vitems, expect = [], []
        for i, ((size, wid, bucket, obj), body) in enumerate(zip(items, out)):
            body = bytearray(body)
            if i % 4 == 1:
                body[int(rng.integers(0, size))] ^= 1 << int(rng.integers(0, 8))
            if i % 7 == 2:
                body = body[:int(rng.integers(0, size))]
            vitems.append((wid, bucket, obj, bytes(body)))
            expect.append(oracle.valid(wid, bucket, obj.Key, obj.Size, obj.WriteCount, bytes(body))[2])
        got = pattern.ValidBatch(vitems)
        assert [None if g is None else str(g) for g in got] == expect
        # large bodies on either engine: generated in parts, validated from a reader in chunks
        pattern.set_stream_chunk(1 << 20)
        try:
            for wid in (6, 7):
                size = (9 << 20) + 512
                obj = Object(oracle.generate_key(wid), size, 3)
                want = oracle.generate(size, wid, "test-bucket", obj.Key, obj.WriteCount, T0)
                with pattern.GenerateParts(size, wid, "test-bucket", obj, 5 << 20) as parts:
                    body = b"".join(parts.part(i).tobytes() for i in range(len(parts)))
                assert body == want
                pattern.ValidStream(wid, "test-bucket", obj, io.BytesIO(body))
                bad = bytearray(body)
                bad[(7 << 20) + 29] ^= 4
                with pytest.raises(pattern.PatternError) as e:
                    pattern.ValidStream(wid, "test-bucket", obj, io.BytesIO(bytes(bad)))
                assert str(e.value) == oracle.valid(wid, "test-bucket", obj.Key, size, obj.WriteCount, bytes(bad))[2]
        finally:
            pattern.set_stream_chunk(32 << 20)
    finally:
        pattern.shutdown()
        pattern.init_engines([0])
    assert pattern.engine_count() == 1
    _check(oracle, items[:5], [_cycle(pattern, it) for it in items[:5]])


def test_calls_in_two_steps_return_what_the_blocking_calls_return(oracle):
    """GenerateBegin / ValidBegin + poll: the job travels in a shared launch while
    the caller is free to do something else; outcomes, error texts included, are
    those of Generate / Valid (held to the oracle).  Two-step calls and blocking
    calls from threads share launches; with micro-batching off a two-step call is
    carried out at once."""
    from oval_amd import pattern
    from oval_amd.object import Object

    pattern.set_clock(T0)
    rng = np.random.default_rng(77)
    items = _objects(rng, 400, oracle)

    def two_step(batch):
        gens = [pattern.GenerateBegin(size, wid, bucket, obj) for size, wid, bucket, obj in batch]
        bodies = [g.result() for g in gens]
        vals, expect = [], []
        for i, ((size, wid, bucket, obj), body) in enumerate(zip(batch, bodies)):
            body = bytearray(body)
            if i % 4 == 1:
                body[int(rng.integers(0, size))] ^= 1 << int(rng.integers(0, 8))
            if i % 7 == 2:
                body = body[:int(rng.integers(0, size))]
            if i % 11 == 3:
                obj = Object(obj.Key, obj.Size, obj.WriteCount + 1)
            vals.append(pattern.ValidBegin(wid, bucket, obj, bytes(body)))
            expect.append(oracle.valid(wid, bucket, obj.Key, obj.Size, obj.WriteCount, bytes(body))[2])
        got = []
        for v in vals:
            try:
                v.result()
                got.append(None)
            except pattern.PatternError as e:
                got.append(str(e))
        return bodies, got, expect

    b0, j0 = pattern.batching_stats()
    pattern.set_direct_callers(0)
    pattern.set_batching(True, window_us=300)
    try:
        bodies, got, expect = two_step(items[:200])
        b1, j1 = pattern.batching_stats()
        assert 200 < j1 - j0 <= 400                      # (a body cut to under one unit has no device work)
        assert b1 - b0 < (j1 - j0) // 4, "two-step calls did not share launches"
        # blocking callers on threads and two-step calls in the same launches
        with ThreadPoolExecutor(12) as ex:
            fut = ex.map(lambda it: _cycle(pattern, it), items[200:320])
            bodies2, got2, expect2 = two_step(items[320:])
            results = list(fut)
        _check(oracle, items[200:320], results)
    finally:
        pattern.set_direct_callers(16)
        pattern.set_batching(False)
    for (size, wid, bucket, obj), body in zip(items[:200] + items[320:], bodies + bodies2):
        assert body == oracle.generate(size, wid, bucket, obj.Key, obj.WriteCount, T0)
    assert got == expect and got2 == expect2
    assert sum(e is not None for e in expect) > 40
    # micro-batching off: Begin does the whole call
    g = pattern.GenerateBegin(items[0][0], items[0][1], items[0][2], items[0][3])
    assert g.poll() and g.result() == bodies[0]
    # argument errors surface at the first poll, as Generate would have returned them
    bad = pattern.GenerateBegin(300, 1, "b", Object("k", 300, 1))
    with pytest.raises(pattern.PatternError, match="generated data size is wrong"):
        bad.result()
    empty = pattern.ValidBegin(1, "b", Object("k", 0, 1), b"")
    assert empty.result() is None


def test_shutdown_does_not_wait_for_ever_for_a_call_nobody_collected(oracle):
    """A call in two steps holds its staging slot until it has been polled to its
    end.  shutdown() waits for such slots for the device time limit (at least a
    second), then leaves them as they are; the call can still be collected
    afterwards, and its slot rejoins the others."""
    import time

    from oval_amd import pattern

    pattern.set_clock(T0)
    rng = np.random.default_rng(91)
    items = _objects(rng, 12, oracle)
    pattern.set_device_timeout(300)
    pattern.set_direct_callers(0)
    pattern.set_batching(True, window_us=100)
    try:
        size, wid, bucket, obj = items[0]
        left = pattern.GenerateBegin(size, wid, bucket, obj)
        t0 = time.monotonic()
        pattern.shutdown()                                      # ends batching too: the job's batch is run on the way
        took = time.monotonic() - t0
        assert 0.9 < took < 5.0, took
        assert left.result() == oracle.generate(size, wid, bucket, obj.Key, obj.WriteCount, T0)
    finally:
        pattern.set_device_timeout(30000)
        pattern.set_direct_callers(16)
        pattern.set_batching(False)
        pattern.shutdown()
        pattern.init_engines([0])
    _check(oracle, items[1:], [_cycle(pattern, it) for it in items[1:]])

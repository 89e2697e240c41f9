"""oval_validate_noalloc (the entry point that allocates nothing) against
oval_layout_create + oval_validate AND against the CPU oracle, through the C
ABI: same result words and counters, bit for bit, on two calls in a row of which
the second is smaller (the static scratch keeps the first call's chunk table),
then on a body with gaps; the size limit; and a table that breaks the ordering
rule, which may give meaningless results but must not write outside the result
and counter arrays it was given."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

T0 = 1709558129112997


def _both(torch, engine, lib, ctx, descs, body):
    """(results, counters) of the layout path and of the no-alloc path."""
    from oval_amd._lib import check
    from oval_amd.batch import span_bytes

    n = len(descs)
    layout = engine.Layout(ctx, descs)
    res_a, cnt_a = layout.validate(body)
    ctx.sync()
    descs_dev = torch.from_numpy(descs.view(np.uint8).reshape(-1).copy()).to("cuda:0")
    res_b = torch.full((max(n, 1), 4), 0x5A5A5A5A, dtype=torch.int32, device="cuda:0")
    cnt_b = torch.zeros(engine.NUM_COUNTERS, dtype=torch.int64, device="cuda:0")
    torch.cuda.synchronize()
    check(lib.oval_validate_noalloc(ctx.handle, descs_dev.data_ptr(), n, span_bytes(descs), body.data_ptr(),
                                    body.numel(), res_b.data_ptr(), cnt_b.data_ptr()))
    ctx.sync()
    a = engine.results_to_numpy(res_a, n)
    b = engine.results_to_numpy(res_b, n)
    return a, cnt_a.cpu().numpy(), b, cnt_b.cpu().numpy()


def _damaged_batch(torch, engine, ctx, n_obj, seed, gap_bytes=0, lo=4096, hi=16384):
    from oval_amd.batch import (WorkerSpace, apply_corruption_torch, build_descs, make_keys, plan_corruption,
                                span_bytes, synthetic_batch)

    descs = synthetic_batch(WorkerSpace(num_obj=n_obj, num_worker=4, start_worker_id=0x561B), range(4),
                            "test-bucket", lo, hi, seed=seed, unix_micro=T0)
    if gap_bytes:
        step = descs["n_units"].astype(np.uint64) * np.uint64(256) + np.uint64(gap_bytes)
        descs["body_offset"][0] = 512
        descs["body_offset"][1:] = 512 + np.cumsum(step[:-1])
    body = torch.zeros(span_bytes(descs), dtype=torch.uint8, device="cuda:0")
    engine.Layout(ctx, descs).fill(body)
    ctx.sync()
    plan = plan_corruption(descs, 2e-3, seed=seed + 1)
    apply_corruption_torch(body, descs, plan)
    torch.cuda.synchronize()
    return descs, body


def test_noalloc_matches_layout_validate_twice_second_smaller(oracle):
    import torch

    from oval_amd import _lib, engine

    lib = _lib.lib()
    ctx = engine.Context.on_torch_stream(0)
    try:
        for n_obj, seed, gap in ((40000, 11, 0), (1200, 12, 0), (3000, 13, 768)):
            descs, body = _damaged_batch(torch, engine, ctx, n_obj, seed, gap)
            a, ca, b, cb = _both(torch, engine, lib, ctx, descs, body)
            assert int(ca[engine_cnt("OBJECTS_BAD")]) > 0
            for f in a.dtype.names:
                bad = np.flatnonzero(a[f] != b[f])
                assert bad.size == 0, (n_obj, f, bad[:5], a[f][bad[:5]], b[f][bad[:5]])
            assert np.array_equal(ca, cb), (n_obj, ca, cb)
            # and both equal the oracle's verdicts over the same damaged bytes
            eres, ecnt = oracle.validate_batch(descs.view(oracle.DESC_DTYPE), body.cpu().numpy())
            for f in eres.dtype.names:
                bad = np.flatnonzero(b[f] != eres[f])
                assert bad.size == 0, (n_obj, f, bad[:5], b[f][bad[:5]], eres[f][bad[:5]])
            assert np.array_equal(cb.astype(np.uint64), ecnt), (n_obj, cb, ecnt)
    finally:
        ctx.close()


def test_noalloc_table_that_breaks_the_rules_stays_inside_its_arrays():
    """The descriptor table of the no-alloc entry point never passes through the
    host, so its ordering cannot be checked: bodies out of order, overlapping,
    beyond the span or absurdly long give results that mean nothing, but every
    write must land in results[0, n_obj) or counters[0, 12) (guard words on both
    sides stay as they were) and the call must end."""
    import torch

    from oval_amd import _lib, engine
    from oval_amd._lib import check
    from oval_amd.batch import span_bytes

    lib = _lib.lib()
    ctx = engine.Context.on_torch_stream(0)
    try:
        descs, body = _damaged_batch(torch, engine, ctx, 400, 31)
        span = span_bytes(descs)
        rng = np.random.default_rng(7)
        broken = descs.copy()
        rng.shuffle(broken)                                     # not ascending any more
        broken["body_offset"][5] = broken["body_offset"][6]     # two objects on the same bytes
        broken["body_offset"][9] = span + (1 << 20)             # beyond the span
        broken["n_units"][11] = 0xFFFFFF00                      # far longer than the body
        broken["body_offset"][13] = (1 << 40) - 256             # unit index near the top of its 32 bits
        n, guard = len(broken), 64
        res = torch.full((n + 2 * guard, 4), 0x5A5A5A5A, dtype=torch.int32, device="cuda:0")
        cnt = torch.full((engine.NUM_COUNTERS + 2 * guard,), 0x1234567, dtype=torch.int64, device="cuda:0")
        cnt[guard:guard + engine.NUM_COUNTERS] = 0
        descs_dev = torch.from_numpy(broken.view(np.uint8).reshape(-1).copy()).to("cuda:0")
        torch.cuda.synchronize()
        check(lib.oval_validate_noalloc(ctx.handle, descs_dev.data_ptr(), n, span, body.data_ptr(), body.numel(),
                                        res[guard:].data_ptr(), cnt[guard:].data_ptr()))
        ctx.sync()
        r, c = res.cpu().numpy(), cnt.cpu().numpy()
        assert (r[:guard] == 0x5A5A5A5A).all() and (r[guard + n:] == 0x5A5A5A5A).all()
        assert (c[:guard] == 0x1234567).all() and (c[guard + engine.NUM_COUNTERS:] == 0x1234567).all()
        assert c[guard + engine_cnt("OBJECTS_CHECKED")] == n
        # the scratch is whole afterwards: a well-formed call right behind it is exact again
        a, ca, b, cb = _both(torch, engine, lib, ctx, descs, body)
        for f in a.dtype.names:
            assert np.array_equal(a[f], b[f]), f
        assert np.array_equal(ca, cb)
    finally:
        ctx.close()


def engine_cnt(name):
    from oval_amd import _lib

    return getattr(_lib, "CNT_" + name)


def test_noalloc_empty_and_limits():
    import torch

    from oval_amd import _lib, engine

    lib = _lib.lib()
    ctx = engine.Context.on_torch_stream(0)
    try:
        cnt = torch.zeros(engine.NUM_COUNTERS, dtype=torch.int64, device="cuda:0")
        body = torch.zeros(4096, dtype=torch.uint8, device="cuda:0")
        res = torch.zeros((4, 4), dtype=torch.int32, device="cuda:0")
        # no objects: nothing checked, nothing bad
        assert lib.oval_validate_noalloc(ctx.handle, None, 0, 0, body.data_ptr(), body.numel(), res.data_ptr(),
                                         cnt.data_ptr()) == 0
        ctx.sync()
        assert not cnt.cpu().numpy().any()
        # beyond the static chunk table: refused, with a message
        rc = lib.oval_validate_noalloc(ctx.handle, body.data_ptr(), 1, (4 << 30) + 256, body.data_ptr(),
                                       (4 << 30) + 256, res.data_ptr(), cnt.data_ptr())
        assert rc == -6 and "static chunk table" in _lib.last_error()
        # body shorter than the span
        rc = lib.oval_validate_noalloc(ctx.handle, body.data_ptr(), 1, 8192, body.data_ptr(), 4096, res.data_ptr(),
                                       cnt.data_ptr())
        assert rc == -6
    finally:
        ctx.close()

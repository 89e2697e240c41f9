"""Objects of one to a few units (the reference's smallest object is a single
256-byte unit, internal/pattern/pattern.go:23-34): the kernels that resolve all
objects of a chunk at once, held to the CPU oracle byte for byte and result for
result, with the selection by average object size at its default, forced on for
every layout, and switched off.  Layouts include what those kernels must hand to
the object walk: runs of zero-unit objects (more than sixteen candidates per
chunk), gaps, parts of objects, names with white space at an edge, worker IDs
outside uint32.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

T0 = 1709558129112997
EVERY = (1 << 64) - 1


@pytest.fixture(scope="module")
def gpu():
    import torch

    from oval_amd import engine

    ctx = engine.Context.on_torch_stream(0)
    yield {"torch": torch, "engine": engine, "ctx": ctx}
    from conftest import session_small_object_bytes

    engine.set_small_object_bytes(*session_small_object_bytes())
    ctx.close()


def _damage(rng, host, exp, descs, units, rounds):
    n = len(descs)
    for _ in range(rounds):
        o = int(rng.integers(0, n))
        if units[o] == 0:
            continue
        u = int(rng.integers(0, units[o]))
        at = int(descs["body_offset"][o]) + u * 256
        kind = int(rng.integers(0, 4))
        if kind == 0:                                          # one bit anywhere in the unit
            host[at + int(rng.integers(0, 256))] ^= np.uint8(1 << int(rng.integers(0, 8)))
        elif kind == 1:                                        # another object's first unit
            o2 = int(rng.integers(0, n))
            if units[o2]:
                src = int(descs["body_offset"][o2])
                host[at:at + 256] = exp[src:src + 256]
        elif kind == 2:                                        # a run of zeros
            host[at:at + int(rng.integers(1, 257))] = 0
        else:                                                  # a stale generation: the write count one lower
            wc = int.from_bytes(host[at + 28:at + 32].tobytes(), "little")
            host[at + 28:at + 32] = np.frombuffer(((wc - 1) & 0xFFFFFFFF).to_bytes(4, "little"), dtype=np.uint8)


def _check(gpu, oracle, descs, units, rng, rounds):
    from test_gpu_parity import _assert_results_equal, _fill, _oracle_fill

    torch = gpu["torch"]
    layout, body = _fill(gpu, descs, extra=512)
    exp = _oracle_fill(oracle, descs, body.numel())
    got = body.cpu().numpy()
    bad = np.flatnonzero(got != exp)
    assert bad.size == 0, f"fill differs from the oracle first at byte {bad[:4]} (unit {bad[:4] // 256})"
    _assert_results_equal(oracle, gpu, layout, body, descs)    # clean
    host = exp.copy()
    _damage(rng, host, exp, descs, units, rounds)
    body.copy_(torch.from_numpy(host))
    got, cnt = _assert_results_equal(oracle, gpu, layout, body, descs)
    layout.close()
    return cnt


def _table(rng, units, gap=0, bucket="test-bucket", worker_ids=None, parts=False):
    from oval_amd.batch import build_descs, make_keys

    n = len(units)
    first = rng.integers(0, 2, size=n) * rng.integers(0, 1 << 24, size=n) if parts else 0
    wid = rng.integers(0, 0x10000, size=n) if worker_ids is None else worker_ids
    return build_descs(bucket, make_keys(rng.integers(0, 1 << 40, size=n, dtype=np.uint64)),
                       rng.integers(0, 1 << 32, size=n), wid,
                       T0 + rng.integers(0, 1 << 40, size=n).astype(np.uint64), np.asarray(units) * 256,
                       first_units=first, gap_bytes=gap)


@pytest.mark.parametrize("threshold", ["default", EVERY, 0])
def test_small_objects_match_the_oracle(gpu, oracle, threshold):
    engine = gpu["engine"]
    engine.set_small_object_bytes(*(engine.SMALL_OBJECT_BYTES_DEFAULT if threshold == "default" else (threshold, threshold)))
    rng = np.random.default_rng(31 if threshold == "default" else 32 + (threshold & 1))
    bad_units = 0
    for trial in range(24):
        n = int(rng.choice([1, 15, 16, 17, 63, 300, 5000, 40000]))
        shape = trial % 6
        if shape == 0:
            units = np.ones(n, dtype=np.int64)                                 # one unit each: sixteen objects per chunk
        elif shape == 1:
            units = rng.integers(1, 4, size=n)
        elif shape == 2:
            units = rng.integers(0, 3, size=n)                                 # zero-unit objects in between
        elif shape == 3:
            units = rng.integers(1, 9, size=n)
            units[rng.integers(0, n, size=max(1, n // 50))] = rng.integers(20, 400)   # a few larger ones among them
        elif shape == 4:
            units = rng.integers(0, 2, size=n)
            for at in rng.integers(0, max(1, n - 40), size=max(1, n // 200)):  # runs of 17-40 zero-unit objects:
                units[at:at + int(rng.integers(17, 41))] = 0                   # beyond the candidates of a chunk
            units[-1] = 1
        else:
            units = rng.integers(1, 5, size=n)
        gap = int(rng.choice([0, 0, 256, 768, 4096 + 256]))
        bucket = str(rng.choice(["test-bucket", "b", " edge-space", "sixteen-chars-ok!"]))
        wid = None
        if trial % 5 == 3:
            wid = rng.integers(0, 0x10000, size=n).astype(np.int64)
            wid[rng.integers(0, n, size=max(1, n // 10))] += 1 << 32           # compared as a Go int: never equal
        descs = _table(rng, units, gap=gap, bucket=bucket, worker_ids=wid, parts=trial % 2 == 1)
        cnt = _check(gpu, oracle, descs, units, rng, rounds=int(rng.integers(0, 30)))
        bad_units += int(cnt[3])
    assert bad_units > 0


def test_a_batch_of_one_unit_objects_across_many_launches(gpu, oracle):
    """1.2 million one- and two-unit objects: several fill launches (the launches
    are cut by the table bytes they depend on), every byte and every result."""
    from oval_amd.batch import apply_corruption_torch, plan_corruption

    from test_gpu_parity import _assert_results_equal, _fill, _oracle_fill

    engine = gpu["engine"]
    engine.set_small_object_bytes(*engine.SMALL_OBJECT_BYTES_DEFAULT)
    rng = np.random.default_rng(7)
    units = rng.integers(1, 3, size=1_200_000)
    descs = _table(rng, units)
    layout, body = _fill(gpu, descs)
    assert layout.launches()[0] > 1
    exp = _oracle_fill(oracle, descs, body.numel())
    assert np.array_equal(body.cpu().numpy(), exp)
    apply_corruption_torch(body, descs, plan_corruption(descs, 1e-3, seed=3))
    gpu["torch"].cuda.synchronize()
    _, cnt = _assert_results_equal(oracle, gpu, layout, body, descs)
    assert cnt[3] > 1000
    layout.close()


def test_selection_follows_the_average_object_size(gpu, oracle):
    """The same layout object reused for small and large objects in turn, with the
    selection at its default: results never depend on which kernels ran."""
    from oval_amd.batch import build_descs, make_keys

    from test_gpu_parity import _assert_results_equal, _oracle_fill

    torch, engine, ctx = gpu["torch"], gpu["engine"], gpu["ctx"]
    engine.set_small_object_bytes(*engine.SMALL_OBJECT_BYTES_DEFAULT)
    rng = np.random.default_rng(11)
    layout = None
    for n, lo, hi in [(3000, 1, 3), (40, 100, 300), (5000, 0, 2), (3, 4000, 5000), (900, 1, 1)]:
        units = rng.integers(lo, hi + 1, size=n)
        descs = build_descs("test-bucket", make_keys(rng.integers(0, 1 << 40, size=n, dtype=np.uint64)),
                            rng.integers(1, 1000, size=n), rng.integers(0, 0x10000, size=n), T0, units * 256)
        if layout is None:
            layout = engine.Layout(ctx, descs)
        else:
            layout.reset(descs)
        body = torch.full((max(int(units.sum()) * 256, 256),), 0xEE, dtype=torch.uint8, device="cuda:0")
        layout.fill(body)
        ctx.sync()
        assert np.array_equal(body.cpu().numpy(), _oracle_fill(oracle, descs, body.numel()))
        _assert_results_equal(oracle, gpu, layout, body, descs)
    layout.close()


def test_eight_million_one_unit_objects_every_byte_and_damage(gpu, oracle):
    """The shape the small-object kernels are for, at benchmark size: 8,000,000
    objects of one unit (2 GB of bodies, 512 MB of descriptors), every byte against
    the oracle on 16 host threads, then 1e-4 of the units damaged and every result
    and counter against the oracle."""
    from oval_amd.batch import WorkerSpace, apply_corruption_torch, plan_corruption, span_bytes, synthetic_batch

    from test_gpu_fullsize import _check_corrupted_batch, _oracle_compare_fill

    torch, engine, ctx = gpu["torch"], gpu["engine"], gpu["ctx"]
    engine.set_small_object_bytes(*engine.SMALL_OBJECT_BYTES_DEFAULT)
    descs = synthetic_batch(WorkerSpace(num_obj=8_000_000, num_worker=25, start_worker_id=0x561B), range(25),
                            "test-bucket", 256, 256, seed=5, unix_micro=T0)
    assert int(descs["n_units"].sum()) == len(descs) == 8_000_000
    layout = engine.Layout(ctx, descs)
    body = torch.empty(span_bytes(descs), dtype=torch.uint8, device="cuda:0")
    layout.fill(body)
    ctx.sync()
    _oracle_compare_fill(oracle, body, descs, 100_000)
    res, cnt = layout.validate(body)
    ctx.sync()
    cnt = cnt.cpu().numpy()
    assert cnt[0] == 8_000_000 and cnt[2] == 8_000_000 and not cnt[1] and not cnt[3] and not cnt[11]
    plan = plan_corruption(descs, 1e-4, seed=9)
    apply_corruption_torch(body, descs, plan)
    torch.cuda.synchronize()
    got, cnt = _check_corrupted_batch(oracle, gpu, layout, body, descs, plan)
    assert cnt[3] > 300
    layout.close()


def test_small_objects_through_the_host_memory_entry_points(gpu, oracle):
    """Bodies that start and end in host memory (oval_fill_batch_host /
    oval_validate_batch_host, 32 MiB slices through the device): 400,000 objects of
    one to three units, i.e. several slices of the small-object kernels, with gaps;
    bytes outside the objects are left alone."""
    from oval_amd.batch import span_bytes

    engine, ctx = gpu["engine"], gpu["ctx"]
    engine.set_small_object_bytes(*engine.SMALL_OBJECT_BYTES_DEFAULT)
    rng = np.random.default_rng(23)
    units = rng.integers(1, 4, size=400_000)
    descs = _table(rng, units, gap=256)
    n = span_bytes(descs)
    assert n > 3 * (32 << 20)
    body = np.full(n, 0xEE, dtype=np.uint8)
    ctx.fill_batch_host(descs, body)
    exp = np.full(n, 0xEE, dtype=np.uint8)
    oracle.fill_batch(descs.view(oracle.DESC_DTYPE), exp)
    assert np.array_equal(body, exp)
    _damage(rng, body, exp, descs, units, rounds=200)
    res, cnt = ctx.validate_batch_host(descs, body)
    eres, ecnt = oracle.validate_batch(descs.view(oracle.DESC_DTYPE), body)
    for f in eres.dtype.names:
        assert np.array_equal(res[f], eres[f]), f
    assert np.array_equal(cnt, ecnt) and ecnt[3] > 50

"""CPU-only checks of the batch tables: key ranges, sharding by worker key
range, synthetic batches, corruption plans (verified with the oracle)."""
import numpy as np
import pytest

from oval_amd.batch import (WorkerSpace, apply_corruption_numpy, build_descs, decide_sizes, make_keys,
                            plan_corruption, shard_workers, span_bytes, synthetic_batch)

T0 = 1709558129112997


def test_make_keys_matches_generate_key(oracle):
    ids = np.array([0, 1, 255, (2 << 24) + 0xE2, (7 << 32) + (200 << 24) + 0xFFFFFF, (1 << 40) - 1], dtype=np.uint64)
    keys = make_keys(ids)
    for i, oid in enumerate(ids):
        assert keys[i].tobytes().decode() == oracle.generate_key(int(oid))
    with pytest.raises(ValueError):
        make_keys(np.array([1 << 40], dtype=np.uint64))


def test_worker_space_matches_readme_key_ranges():
    # README.md:71-74: 4 workers x 256 objects, runner 0
    sp = WorkerSpace(num_obj=1024, num_worker=4, runner_id=0, start_worker_id=0x614)
    for w, (head, tail) in enumerate([("ov0000000000", "ov00000000ff"), ("ov0001000000", "ov00010000ff"),
                                      ("ov0002000000", "ov00020000ff"), ("ov0003000000", "ov00030000ff")]):
        keys = make_keys(sp.obj_ids([w]))
        assert keys[0].tobytes().decode() == head and keys[-1].tobytes().decode() == tail
    assert list(np.unique(sp.stamped_worker_ids(range(4)))) == [0x614, 0x615, 0x616, 0x617]
    sp2 = WorkerSpace(1024, 4, start_worker_id=0xFFFE)
    assert list(sp2.stamped_worker_ids(range(4))[::256]) == [0xFFFE, 0xFFFF, 0, 1]     # runner.go:113 wraps


@pytest.mark.parametrize("nw,ws", [(64, 8), (64, 1), (7, 2), (255, 8), (4, 8), (3, 4)])
def test_shard_workers_partitions(nw, ws):
    got = [w for r in range(ws) for w in shard_workers(nw, ws, r)]
    assert got == list(range(nw))
    sizes = [len(shard_workers(nw, ws, r)) for r in range(ws)]
    assert max(sizes) - min(sizes) <= 1


def test_synthetic_batch_is_shard_invariant():
    sp = WorkerSpace(num_obj=4096, num_worker=8, start_worker_id=77)
    whole = synthetic_batch(sp, range(8), "test-bucket", 4096, 16384, seed=5, unix_micro=T0)
    parts = [synthetic_batch(sp, shard_workers(8, 4, r), "test-bucket", 4096, 16384, seed=5, unix_micro=T0)
             for r in range(4)]
    cat = np.concatenate(parts)
    for f in ("key", "write_count", "worker_id", "n_units", "bucket", "unix_micro"):
        assert (cat[f] == whole[f]).all(), f
    assert all(p["body_offset"][0] == 0 for p in parts)       # each rank's buffer starts at 0
    assert whole["n_units"].min() >= 16 and whole["n_units"].max() <= 64
    assert 1 <= whole["write_count"].min() and whole["write_count"].max() <= 1000


def test_decide_sizes_matches_scalar(oracle):
    rng = np.random.default_rng(0)
    y = rng.random(5000)
    for lo, hi in ((4096, 16384), (4096, 1 << 20), (256, 256)):
        v = decide_sizes(y, lo, hi)
        s = np.array([oracle.decide_size(lo, hi, float(t)) for t in y])
        assert (np.abs(v - s) <= 256).all() and (v == s).mean() > 0.999
        assert v.min() >= lo and v.max() <= hi


def test_build_descs_layout_and_gaps(oracle):
    d = build_descs("test-long-long-long-bucket", ["a", "bb", "ccc"], [1, 2, 3], 9, T0, [512, 0, 1024], gap_bytes=256)
    assert list(d["body_offset"]) == [0, 768, 1024] and list(d["n_units"]) == [2, 0, 4]
    assert bytes(d[0]["bucket"]) == b"test-long-long-l" and d[0]["bucket_len"] == 16
    assert span_bytes(d) == 2048
    body = np.full(2048, 0xEE, dtype=np.uint8)
    oracle.fill_batch(d.view(oracle.DESC_DTYPE), body)
    assert (body[512:768] == 0xEE).all() and (body[768:1024] == 0xEE).all()     # gaps untouched
    assert body[0:16].tobytes() == b"test-long-long-l"


def test_corruption_plan_agrees_with_oracle(oracle):
    sp = WorkerSpace(num_obj=512, num_worker=4, start_worker_id=9)
    d = synthetic_batch(sp, range(4), "test-bucket", 4096, 16384, seed=1, unix_micro=T0)
    body = np.zeros(span_bytes(d), dtype=np.uint8)
    D = d.view(oracle.DESC_DTYPE)
    oracle.fill_batch(D, body)
    plan = plan_corruption(d, fraction=2e-3, seed=3)
    assert len(plan.flip_byte) + len(plan.stale_obj) >= 30
    apply_corruption_numpy(body, d, plan)
    res, cnt = oracle.validate_batch(D, body)
    bad = np.flatnonzero(res["first_bad_unit"] != oracle.NO_BAD)
    # every stale-generation unit is reference-visible (write count), flips only in bytes [0,40)
    assert set(plan.stale_obj) <= set(bad)
    vis = plan._flip_obj[(plan.flip_byte % 256) < 40]
    assert set(bad) == set(plan.stale_obj) | set(vis)
    # strict sees everything except flips inside the time bytes
    strict_bad = np.flatnonzero(res["strict_first_bad_unit"] != oracle.NO_BAD)
    hidden = (plan.flip_byte % 256 >= 40) & (plan.flip_byte % 256 < 48)
    assert set(strict_bad) == set(plan.stale_obj) | set(plan._flip_obj[~hidden])
    # first bad unit of a stale-only object is its lowest stale unit, fields = write count
    for o in set(plan.stale_obj) - set(plan._flip_obj):
        assert res["first_bad_unit"][o] == plan.stale_unit[plan.stale_obj == o].min()
        assert res["first_bad_fields"][o] == 4
    assert cnt[1] == len(bad) and cnt[3] == res["bad_units"].sum()


def test_split_batches_rebases_and_covers():
    from oval_amd.batch import split_batches, span_bytes
    sizes = np.array([4096, 8192, 256, 65536, 4096, 1 << 20, 512, 512], dtype=np.int64)
    d = build_descs("b", make_keys(np.arange(len(sizes))), 1, 2, 3, sizes)
    parts = split_batches(d, 70000)
    assert sum(len(p) for p in parts) == len(d)
    assert [int(p["body_offset"][0]) for p in parts] == [0] * len(parts)
    for p in parts:
        assert span_bytes(p) <= 70000 or len(p) == 1            # the 1 MiB object stands alone
    cat = np.concatenate(parts)
    for f in ("bucket", "key", "write_count", "worker_id", "n_units", "first_unit"):
        assert (cat[f] == d[f]).all()
    # offsets inside a part keep the original spacing
    k = 0
    for p in parts:
        assert (p["body_offset"] == d["body_offset"][k:k + len(p)] - d["body_offset"][k]).all()
        k += len(p)
    assert split_batches(d[:0], 100) == []
    with pytest.raises(ValueError):
        split_batches(d, 0)


def test_bench_shards_one_key_space_by_worker_range():
    """bench.py --sharding workers: the ranks' batches are disjoint worker ranges
    of ONE key space (process ID 0) and together cover it (runner.go:117-124)."""
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    world = 8
    shards = [bench.build_batch("cfg5", r, world, "workers", 2500, 1)[0] for r in range(world)]
    keys = np.concatenate([s["key"] for s in shards])
    assert len(np.unique(keys, axis=0)) == len(keys) == 8 * 2500
    texts = [bytes(k).decode() for k in keys]
    assert all(t.startswith("ov00") for t in texts)                      # process ID 0 everywhere
    workers = sorted({int(t[4:6], 16) for t in texts})
    assert workers == list(range(200))                                   # 25 workers per rank, 200 < 256
    for r, s in enumerate(shards):
        w = {int(bytes(k).decode()[4:6], 16) for k in s["key"]}
        assert w == set(range(25 * r, 25 * r + 25))
    # weak scaling means the same work per GPU at every N: same objects, workers and (per worker
    # stream) sizes on rank 0 whether the job has 1, 2, 4 or 8 ranks
    for name in ("cfg2", "cfg3", "cfg5"):
        n_obj = 5000 if name != "cfg3" else 0
        ref = bench.build_batch(name, 0, 1, "workers", n_obj, 1)[0]
        for w in (2, 4, 8):
            d = bench.build_batch(name, 0, w, "workers", n_obj, 1)[0]
            assert len(d) == len(ref) and np.array_equal(d["n_units"], ref["n_units"])
            assert np.array_equal(d["key"], ref["key"])
    full = bench.WORKLOADS
    assert full["cfg2"][0] % full["cfg2"][1] == 0 and full["cfg5"][0] % full["cfg5"][1] == 0
    assert 8 * full["cfg5"][0] == 10_000_000 and 8 * full["cfg5"][1] < 256
    # the same objects whatever the world size: a worker's stream does not depend on the cut
    one = bench.build_batch("cfg5", 0, 1, "workers", 0, 1)[1]
    assert "one key space" in one
    by_process = [bench.build_batch("cfg2", r, 2, "process", 640, 1)[0] for r in range(2)]
    assert bytes(by_process[1]["key"][0]).decode().startswith("ov01")     # process ID = rank


def test_build_descs_flags_what_the_fast_compare_would_miss(oracle):
    """Names with white space at an edge and worker IDs outside uint32 can never
    validate in the reference (pattern.go:140,147,166-167): the table says so, as
    oval_desc_init would, and the oracle reports every unit of such objects."""
    from oval_amd._lib import DESC_BUCKET_SLOW, DESC_KEY_SLOW, DESC_WORKER_OOR
    from oval_amd.batch import build_descs

    d = build_descs("b", ["plain", " lead", "trail\t", "ok"], [1, 1, 1, 1], [7, 7, 7, -1], 5, 512)
    assert list(d["flags"]) == [0, DESC_KEY_SLOW, DESC_KEY_SLOW, DESC_WORKER_OOR]
    d2 = build_descs(" b", ["k"], [1], [1 << 32], 5, 256)
    assert d2["flags"][0] == DESC_BUCKET_SLOW | DESC_WORKER_OOR
    body = np.zeros(int(d["n_units"].sum()) * 256 + 0, dtype=np.uint8)
    dd = d.copy()
    oracle.fill_batch(dd.view(oracle.DESC_DTYPE), body)
    res, cnt = oracle.validate_batch(dd.view(oracle.DESC_DTYPE), body)
    assert list(res["bad_units"]) == [0, 2, 2, 2]
    with pytest.raises(ValueError):
        build_descs("b", np.full((1, 12), 0xC3, dtype=np.uint8), [1], [1], 5, 256)


@pytest.mark.parametrize("seed,fraction", [(1, 1e-3), (2, 2e-2), (3, 0.3)])
def test_expected_counters_from_the_plan_equal_the_oracle(oracle, seed, fraction):
    """The plan-derived counters (used to check batches too large for any CPU
    pass) agree with the oracle run over the damaged bytes."""
    from oval_amd.batch import WorkerSpace, expected_counters, synthetic_batch

    d = synthetic_batch(WorkerSpace(600, 3, start_worker_id=9), range(3), "test-bucket", 4096, 65536, seed=seed, unix_micro=77)
    plan = plan_corruption(d, fraction, seed=seed)
    body = np.zeros(span_bytes(d), dtype=np.uint8)
    oracle.fill_batch(d.view(oracle.DESC_DTYPE), body)
    apply_corruption_numpy(body, d, plan)
    _, cnt = oracle.validate_batch(d.view(oracle.DESC_DTYPE), body)
    assert np.array_equal(expected_counters(d, plan), cnt), (expected_counters(d, plan), cnt)

"""Multipart-sized bodies (internal/s3client/s3client.go:164-236 uploads bodies of
100 MiB and more in parts): the validate kernel that reads with one load per
thread from workgroups of 1024 threads and looks its object up once per
workgroup, held to the CPU oracle result for result, with the selection by
average object size at its default and forced on for every layout.  Layouts
include what sends a workgroup or a wavefront to the walk over its own four
units: object boundaries inside a workgroup's 16 KiB, gaps, objects smaller than
a workgroup's range, flagged names, worker IDs outside uint32, damage next to
every kind of seam.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

T0 = 1709558129112997


@pytest.fixture(scope="module")
def gpu():
    import torch

    from oval_amd import engine

    ctx = engine.Context.on_torch_stream(0)
    yield {"torch": torch, "engine": engine, "ctx": ctx}
    from conftest import session_big_object_bytes

    engine.set_big_object_bytes(session_big_object_bytes())
    ctx.close()


def _table(rng, units, gap=0, bucket="test-bucket", worker_ids=None, parts=False):
    from oval_amd.batch import build_descs, make_keys

    n = len(units)
    first = rng.integers(0, 2, size=n) * rng.integers(0, 1 << 22, size=n) if parts else 0
    wid = rng.integers(0, 0x10000, size=n) if worker_ids is None else worker_ids
    return build_descs(bucket, make_keys(rng.integers(0, 1 << 40, size=n, dtype=np.uint64)),
                       rng.integers(0, 1 << 32, size=n), wid,
                       T0 + rng.integers(0, 1 << 40, size=n).astype(np.uint64), np.asarray(units) * 256,
                       first_units=first, gap_bytes=gap)


def _damage_at_seams(rng, host, exp, descs, units):
    """Single bits, stale write counts and zeroed runs: at random places, in the
    first and last unit of objects, and on either side of 1 KiB, 4 KiB and 16 KiB
    seams of the body."""
    n = len(descs)
    spots = []
    for o in rng.integers(0, n, size=min(n, 12)):
        if units[o] == 0:
            continue
        base = int(descs["body_offset"][o]) // 256
        spots += [base, base + int(units[o]) - 1, base + int(rng.integers(0, units[o]))]
        for seam in (4, 16, 64):
            u = (base + int(rng.integers(0, units[o]))) // seam * seam
            spots += [u - 1, u]
    ends = (descs["body_offset"] // 256 + descs["n_units"]).astype(np.int64)
    starts = (descs["body_offset"] // 256).astype(np.int64)
    for u in spots:
        o = int(np.searchsorted(ends, u, side="right"))
        if o >= n or u < starts[o]:
            continue                                             # a unit in a gap belongs to nobody
        at = u * 256
        kind = int(rng.integers(0, 3))
        if kind == 0:
            host[at + int(rng.integers(0, 256))] ^= np.uint8(1 << int(rng.integers(0, 8)))
        elif kind == 1:
            wc = int.from_bytes(host[at + 28:at + 32].tobytes(), "little")
            host[at + 28:at + 32] = np.frombuffer(((wc - 1) & 0xFFFFFFFF).to_bytes(4, "little"), dtype=np.uint8)
        else:
            host[at:at + int(rng.integers(1, 257))] = 0


def _check(gpu, oracle, descs, units, rng):
    from test_gpu_parity import _assert_results_equal, _fill, _oracle_fill

    torch = gpu["torch"]
    layout, body = _fill(gpu, descs, extra=512)
    exp = _oracle_fill(oracle, descs, body.numel())
    assert np.array_equal(body.cpu().numpy(), exp)
    _assert_results_equal(oracle, gpu, layout, body, descs)      # clean (or every unit flagged)
    host = exp.copy()
    _damage_at_seams(rng, host, exp, descs, units)
    body.copy_(torch.from_numpy(host))
    _, cnt = _assert_results_equal(oracle, gpu, layout, body, descs)
    layout.close()
    return cnt


@pytest.mark.parametrize("threshold", ["default", 0])
def test_big_objects_match_the_oracle(gpu, oracle, threshold):
    engine = gpu["engine"]
    engine.set_big_object_bytes(engine.BIG_OBJECT_BYTES_DEFAULT if threshold == "default" else threshold)
    rng = np.random.default_rng(41 + (threshold == 0))
    bad = 0
    for trial in range(10):
        shape = trial % 5
        if shape == 0:
            units = rng.integers(4096, 40000, size=6)             # 1-10 MiB, boundaries anywhere
        elif shape == 1:
            units = rng.integers(4096, 9000, size=8) // 64 * 64   # boundaries on workgroup seams
        elif shape == 2:
            units = np.concatenate([rng.integers(8192, 30000, size=3), rng.integers(1, 70, size=20),
                                    rng.integers(8192, 30000, size=3)])   # objects smaller than a workgroup's range among them
            rng.shuffle(units)
        elif shape == 3:
            units = rng.integers(5000, 20000, size=5)
            units[2] = 0                                           # a zero-unit object
        else:
            units = np.array([160000])                            # one 40 MiB object
        gap = int(rng.choice([0, 0, 256, 4096 + 256, 16384]))
        bucket = "test-bucket" if trial % 4 else " edge-space"
        wid = None
        if trial % 3 == 2:
            wid = rng.integers(0, 0x10000, size=len(units)).astype(np.int64)
            wid[0] += 1 << 32
        descs = _table(rng, units, gap=gap, bucket=bucket, worker_ids=wid, parts=trial % 2 == 1)
        cnt = _check(gpu, oracle, descs, units, rng)
        bad += int(cnt[3])
    assert bad > 0

"""world_size-2 rehearsal of the multi-GPU path on CPU (gloo): objects shard by
worker key range, every rank validates only its own shard, and one all-reduce
of the counter vector gives every rank the batch totals.  The per-shard compute
is done by the oracle here (there is no GPU in this test)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oval_amd.batch import WorkerSpace, apply_corruption_numpy, plan_corruption, span_bytes, synthetic_batch
from oval_amd.dist import all_reduce_counters, all_reduce_first_bad, shard_batch

T0 = 1709558129112997
SPACE = dict(num_obj=768, num_worker=6, runner_id=1, start_worker_id=0xFFFD)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _shard_results(rank, world):
    from oracle import oracle as O

    sp = WorkerSpace(**SPACE)
    d = shard_batch(sp, world, rank, "test-bucket", 4096, 16384, seed=7, unix_micro=T0)
    body = np.zeros(span_bytes(d), dtype=np.uint8)
    O.fill_batch(d.view(O.DESC_DTYPE), body)
    apply_corruption_numpy(body, d, plan_corruption(d, 1e-3, seed=100 + rank))
    return O.validate_batch(d.view(O.DESC_DTYPE), body)[0]["first_bad_unit"]


def _shard_counters(rank, world):
    from oracle import oracle as O

    sp = WorkerSpace(**SPACE)
    d = shard_batch(sp, world, rank, "test-bucket", 4096, 16384, seed=7, unix_micro=T0)
    body = np.zeros(span_bytes(d), dtype=np.uint8)
    O.fill_batch(d.view(O.DESC_DTYPE), body)
    apply_corruption_numpy(body, d, plan_corruption(d, 1e-3, seed=100 + rank))
    _, cnt = O.validate_batch(d.view(O.DESC_DTYPE), body)
    return d, cnt


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d, cnt = _shard_counters(rank, world)
    t = torch.from_numpy(cnt.astype(np.int64))
    all_reduce_counters(t)
    np.save(os.path.join(out_dir, f"r{rank}.npy"), t.numpy())
    np.save(os.path.join(out_dir, f"k{rank}.npy"), d["key"])
    # the optional second form: the first-bad-unit table of the whole job on every rank
    per = SPACE["num_obj"] // world
    table = torch.full((SPACE["num_obj"],), -1, dtype=torch.int32)
    table[rank * per:(rank + 1) * per] = torch.from_numpy(_shard_results(rank, world).view(np.int32))
    all_reduce_first_bad(table)
    np.save(os.path.join(out_dir, f"t{rank}.npy"), table.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_sum_to_the_whole(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    got = [np.load(tmp_path / f"r{r}.npy") for r in range(world)]
    assert (got[0] == got[1]).all()                       # every rank holds the totals
    expect = sum(_shard_counters(r, world)[1].astype(np.int64) for r in range(world))
    assert (got[0] == expect).all()
    assert got[0][0] == SPACE["num_obj"] and got[0][1] > 0 and got[0][3] >= got[0][1]
    tables = [np.load(tmp_path / f"t{r}.npy").view(np.uint32) for r in range(world)]
    whole_table = np.concatenate([_shard_results(r, world) for r in range(world)])
    assert (tables[0] == whole_table).all() and (tables[1] == whole_table).all()
    assert (whole_table != 0xFFFFFFFF).sum() == got[0][1]
    # the shards are disjoint and cover every key of the run
    keys = np.concatenate([np.load(tmp_path / f"k{r}.npy") for r in range(world)])
    whole = synthetic_batch(WorkerSpace(**SPACE), range(SPACE["num_worker"]), "test-bucket", 4096, 16384, 7, T0)
    assert (keys == whole["key"]).all() and len(np.unique(keys, axis=0)) == SPACE["num_obj"]


def test_bench_starts_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with no WORLD_SIZE around it: the parent starts two
    fresh rank processes before anything touches a GPU, relays rank 0's one line
    and fails when a rank fails.  --rehearse-launch keeps the engine out of it (no
    GPU here): rendezvous, the cut of the key space and one sum over ranks."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--rehearse-launch", "--objects", "5000"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=240)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] is None and "rehearsal" in d
    assert d["config"]["ranks_seen"] == 2 and d["config"]["launcher"] == "self"
    assert d["config"]["objects_all"] == 2 * d["config"]["objects_per_gpu"] == 10000
    # a rank that fails (no GPU in this container: the engine refuses to start) fails the whole job
    if not _has_gpu():
        p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--backend", "gloo", "--objects", "100"],
                           cwd=root, env=env, capture_output=True, text=True, timeout=240)
        assert p.returncode != 0 and p.stdout.strip() == ""
        assert "needs a GPU" in p.stderr and "a rank failed" in p.stderr


def _has_gpu():
    import torch

    return torch.cuda.is_available()

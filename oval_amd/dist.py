"""Multi-GPU glue: one process per GPU, objects sharded by worker key range,
and ONE all-reduce of the batch counters (RCCL over xGMI when the backend is
"nccl"; "gloo" in CPU rehearsals).  No body byte crosses GPUs."""
from __future__ import annotations

import os

from .batch import WorkerSpace, shard_workers, synthetic_batch


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_process_group(backend: str):
    """Rendezvous over MASTER_ADDR/MASTER_PORT (127.0.0.1 on one node)."""
    import torch.distributed as dist

    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        rank, world = env_rank_world()
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return dist


def shard_batch(space: WorkerSpace, world_size: int, rank: int, bucket: str, min_size: int, max_size: int,
                seed: int, unix_micro: int):
    """Descriptors of the objects this rank owns."""
    workers = shard_workers(space.num_worker, world_size, rank)
    return synthetic_batch(space, workers, bucket, min_size, max_size, seed, unix_micro)


def all_reduce(t, op=None):
    """dist.all_reduce in place; a device tensor under the gloo backend (CPU
    rehearsal of the multi-GPU path on a box with fewer GPUs than ranks) goes
    through host memory."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return t
    op = dist.ReduceOp.SUM if op is None else op
    if t.is_cuda and dist.get_backend() == "gloo":
        host = t.cpu()
        dist.all_reduce(host, op=op)
        t.copy_(host)
    else:
        dist.all_reduce(t, op=op)
    return t


def all_reduce_counters(counters):
    """Sums the (12,) int64 counter vector over all ranks, in place: the single
    collective of a batch."""
    return all_reduce(counters)


def all_reduce_first_bad(table):
    """The optional second form of the exchange: `table` holds one uint32 first-bad
    unit per object of the WHOLE job (0xFFFFFFFF = none; this rank's objects filled
    in, everything else none), as an int32 tensor; the minimum over all ranks, in
    place, gives every rank the table of all objects.  torch has no unsigned
    minimum, so the sign bit is flipped around the reduction (unsigned order =
    signed order of the flipped values)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return table
    flipped = table ^ torch.iinfo(torch.int32).min
    all_reduce(flipped, dist.ReduceOp.MIN)
    table.copy_(flipped ^ torch.iinfo(torch.int32).min)
    return table

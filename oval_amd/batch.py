"""Descriptor tables for whole batches of objects.

Host-side bookkeeping only (numpy): which objects a batch holds, where their
bodies lie, how they shard over GPUs, and where a test wants corruption
injected.  The key ranges and stamped worker IDs follow the reference's runner
(internal/runner/runner.go:106-125) and object package
(internal/object/object.go:48-69).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from ._lib import DESC_BUCKET_SLOW, DESC_DTYPE, DESC_KEY_SLOW, DESC_WORKER_OOR, UNIT

_HEX = np.frombuffer(b"0123456789abcdef", dtype=np.uint8)
MAX_WORKER_ID = 0x10000  # runner.go:22-24


def make_keys(obj_ids: np.ndarray) -> np.ndarray:
    """object.generateKey for an array of non-negative IDs below 2^40:
    "ov" + 10 lower-case hex digits (object.go:48-50) -> (n, 12) uint8."""
    ids = np.asarray(obj_ids, dtype=np.uint64)
    if ids.size and int(ids.max()) >= (1 << 40):
        raise ValueError("object IDs of 2^40 and above give keys longer than 12 bytes")
    out = np.empty((ids.size, 12), dtype=np.uint8)
    out[:, 0] = ord("o")
    out[:, 1] = ord("v")
    for k in range(10):
        out[:, 2 + k] = _HEX[((ids >> np.uint64(4 * (9 - k))) & np.uint64(15)).astype(np.intp)]
    return out


def decide_sizes(y: np.ndarray, min_size: int, max_size: int) -> np.ndarray:
    """pattern.DecideSize (pattern.go:63-70) applied to an array of uniform
    samples y in [0,1).  Used to shape synthetic batches; a size that falls on
    a rounding boundary may differ by one unit from the scalar function."""
    if min_size < UNIT or min_size % UNIT or max_size % UNIT or max_size < min_size:
        raise ValueError("sizes must be multiples of 256 with 256 <= min <= max")
    x = -np.log2(1.0 - 0.5 * np.asarray(y, dtype=np.float64))
    steps = (float((max_size - min_size) // UNIT + 1) * x).astype(np.int64)
    return min_size + UNIT * steps


def pad_name(name: bytes, width: int) -> np.ndarray:
    """Left-justified, space-padded ASCII field ("%-<width>s", pattern.go:95)."""
    if len(name) > width:
        raise ValueError(f"name {name!r} is longer than {width} bytes")
    if any(c >= 0x80 for c in name):
        raise ValueError("batch tables take ASCII names only")
    return np.frombuffer(name + b" " * (width - len(name)), dtype=np.uint8)


def _edge_space(field: np.ndarray, lens: np.ndarray) -> np.ndarray:
    """Per row: the ASCII name of length lens[i] starts or ends with white space."""
    field = np.asarray(field)
    first = field[:, 0]
    last = field[np.arange(len(lens)), np.maximum(lens.astype(np.int64) - 1, 0)]
    return (lens > 0) & (((first >= 9) & (first <= 13)) | (first == 32) | ((last >= 9) & (last <= 13)) | (last == 32))


def build_descs(bucket: str | bytes, keys: np.ndarray, write_counts, worker_ids, unix_micro,
                sizes_bytes, first_units=0, gap_bytes: int = 0) -> np.ndarray:
    """One descriptor per object; bodies are laid out back to back (plus
    `gap_bytes` between objects) in the order given.

    bucket       one name for the whole table, cut to 16 bytes like pattern.go:74-76
    keys         (n, 12) uint8 from make_keys, or a list of str/bytes (<= 12 ASCII bytes)
    sizes_bytes  object sizes; only whole units count (Size / 256, pattern.go:79,123)
    """
    b = bucket.encode() if isinstance(bucket, str) else bytes(bucket)
    b = b[:16]
    if isinstance(keys, np.ndarray) and keys.dtype == np.uint8 and keys.ndim == 2:
        n = keys.shape[0]
        key_field = keys
        key_len = np.full(n, 12, dtype=np.uint8)
    else:
        ks = [k.encode() if isinstance(k, str) else bytes(k) for k in keys]
        n = len(ks)
        key_field = np.stack([pad_name(k, 12) for k in ks]) if n else np.empty((0, 12), np.uint8)
        key_len = np.array([len(k) for k in ks], dtype=np.uint8)
    d = np.zeros(n, dtype=DESC_DTYPE)
    d["bucket"] = pad_name(b, 16)
    d["bucket_len"] = len(b)
    d["key"] = key_field
    d["key_len"] = key_len
    if (np.asarray(key_field) >= 0x80).any():
        raise ValueError("batch tables take ASCII names only")
    d["write_count"] = np.asarray(write_counts, dtype=np.int64).astype(np.uint32)
    wid = np.asarray(worker_ids, dtype=np.int64)
    d["worker_id"] = wid.astype(np.uint32)
    # validDataUnit compares the worker ID as a Go int (pattern.go:166-167): one outside uint32
    # can never equal what a unit holds, which the descriptor has to say (oval_desc_init does too)
    d["flags"] = np.where((wid < 0) | (wid > 0xFFFFFFFF), np.uint16(DESC_WORKER_OOR), np.uint16(0))
    # names with white space at an edge can never equal their trimmed field (pattern.go:140,147);
    # the library derives these two bits itself, they are set here so that the table also reads right
    if n:
        blen = np.full(n, len(b), dtype=np.uint8)
        d["flags"] |= np.where(_edge_space(np.broadcast_to(pad_name(b, 16), (n, 16)), blen), np.uint16(DESC_BUCKET_SLOW), np.uint16(0))
        d["flags"] |= np.where(_edge_space(key_field, key_len), np.uint16(DESC_KEY_SLOW), np.uint16(0))
    d["unix_micro"] = np.asarray(unix_micro, dtype=np.uint64)
    d["first_unit"] = np.asarray(first_units, dtype=np.uint32)
    units = (np.asarray(sizes_bytes, dtype=np.int64) // UNIT).astype(np.uint64)
    units = np.broadcast_to(units, (n,)) if units.ndim == 0 else units
    d["n_units"] = units.astype(np.uint32)
    step = units * np.uint64(UNIT) + np.uint64(gap_bytes)
    offs = np.zeros(n, dtype=np.uint64)
    if n > 1:
        np.cumsum(step[:-1], out=offs[1:])
    d["body_offset"] = offs
    return d


def split_batches(descs: np.ndarray, max_bytes: int) -> list:
    """Cuts a descriptor table (bodies back to back, ascending offsets) into
    consecutive sub-batches whose bodies span at most `max_bytes` each, with
    offsets rebased to 0: the pieces a body buffer of that size is reused for.
    An object larger than max_bytes gets a sub-batch of its own."""
    if len(descs) == 0:
        return []
    if max_bytes <= 0:
        raise ValueError("max_bytes must be positive")
    starts = descs["body_offset"].astype(np.uint64)
    ends = starts + descs["n_units"].astype(np.uint64) * np.uint64(UNIT)
    out, i, n = [], 0, len(descs)
    while i < n:
        base = starts[i]
        # last object whose end still fits; ends[] ascends because the table does
        j = int(np.searchsorted(ends, base + np.uint64(max_bytes), side="right"))
        j = max(j, i + 1)
        part = descs[i:j].copy()
        part["body_offset"] -= base
        out.append(part)
        i = j
    return out


def span_bytes(descs: np.ndarray) -> int:
    if len(descs) == 0:
        return 0
    return int((descs["body_offset"] + descs["n_units"].astype(np.uint64) * np.uint64(UNIT)).max())


# ---------------------------------------------------------------------------
# worker key ranges and sharding (runner.go:106-125)
# ---------------------------------------------------------------------------
@dataclass
class WorkerSpace:
    """The key space of one run: `num_worker` workers, each owning
    num_obj // num_worker objects with IDs (runner_id << 32) + (i << 24) + j."""
    num_obj: int
    num_worker: int
    runner_id: int = 0
    start_worker_id: int = 0

    def __post_init__(self):
        if not (0 < self.num_worker < 256):          # cmd/root.go:170-173
            raise ValueError("num_worker must be in [1, 255]")
        if self.num_obj < self.num_worker or self.num_obj > 0x1000000 * self.num_worker:
            raise ValueError("num_obj out of range for num_worker")
        if self.per_worker > 0x1000000:
            raise ValueError("more than 2^24 objects per worker do not fit the key's 6 hex digits")

    @property
    def per_worker(self) -> int:
        return self.num_obj // self.num_worker

    def obj_ids(self, workers) -> np.ndarray:
        w = np.asarray(list(workers), dtype=np.uint64)
        j = np.arange(self.per_worker, dtype=np.uint64)
        base = (np.uint64(self.runner_id) << np.uint64(32)) + (w << np.uint64(24))
        return (base[:, None] + j[None, :]).reshape(-1)

    def stamped_worker_ids(self, workers) -> np.ndarray:
        w = np.asarray(list(workers), dtype=np.int64)
        ids = (self.start_worker_id + w) % MAX_WORKER_ID          # runner.go:113
        return np.repeat(ids, self.per_worker)


def shard_workers(num_worker: int, world_size: int, rank: int) -> range:
    """Contiguous range of worker indices owned by `rank`: objects shard by
    worker key range, so no body byte ever needs to cross GPUs."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    base, rem = divmod(num_worker, world_size)
    lo = rank * base + min(rank, rem)
    return range(lo, lo + base + (1 if rank < rem else 0))


def synthetic_batch(space: WorkerSpace, workers, bucket: str, min_size: int, max_size: int, seed: int,
                    unix_micro: int, max_write_count: int = 1000) -> np.ndarray:
    """Descriptors of every object of the given workers: sizes by DecideSize,
    write counts uniform in [1, max_write_count], one pinned time."""
    workers = list(workers)
    ids = space.obj_ids(workers)
    n = ids.size
    sizes = np.empty(n, dtype=np.int64)
    wcs = np.empty(n, dtype=np.int64)
    # one independent stream per worker, so a worker's objects do not depend on
    # how workers are sharded over ranks
    for wi, w in enumerate(workers):
        rng = np.random.Generator(np.random.PCG64([seed, space.runner_id, int(w)]))
        sl = slice(wi * space.per_worker, (wi + 1) * space.per_worker)
        sizes[sl] = decide_sizes(rng.random(space.per_worker), min_size, max_size)
        wcs[sl] = rng.integers(1, max_write_count + 1, size=space.per_worker)
    return build_descs(bucket, make_keys(ids), wcs, space.stamped_worker_ids(workers), unix_micro, sizes)


# ---------------------------------------------------------------------------
# corruption injection (test and benchmark inputs; the reference only detects)
# ---------------------------------------------------------------------------
@dataclass
class CorruptionPlan:
    flip_byte: np.ndarray     # absolute byte offsets in the batch buffer
    flip_mask: np.ndarray     # uint8, one bit set
    stale_obj: np.ndarray     # object index of each stale-generation unit
    stale_unit: np.ndarray    # unit index inside that object
    stale_back: np.ndarray    # generations to go back (>= 1)

    @property
    def touched_objects(self) -> np.ndarray:
        return np.unique(np.concatenate([self._flip_obj, self.stale_obj]))

    _flip_obj: np.ndarray = None  # filled by plan_corruption


def plan_corruption(descs: np.ndarray, fraction: float, seed: int, min_units: int = 2) -> CorruptionPlan:
    """Chooses ~fraction of all units (at least min_units): half get one random
    bit flipped anywhere in their 256 bytes, half are replaced by the same unit
    of an earlier generation (write count - k, older time)."""
    rng = np.random.Generator(np.random.PCG64([seed, 0xC0FFEE]))
    n_units = descs["n_units"].astype(np.int64)
    total = int(n_units.sum())
    k = min(total, max(min_units, int(round(total * fraction))))
    chosen = np.unique(rng.integers(0, max(total, 1), size=k)) if total else np.empty(0, np.int64)
    ends = np.cumsum(n_units)
    obj = np.searchsorted(ends, chosen, side="right")
    unit = chosen - (ends[obj] - n_units[obj])
    half = len(chosen) // 2
    perm = rng.permutation(len(chosen))
    f, s = perm[:half], perm[half:]
    unit_off = descs["body_offset"][obj].astype(np.int64) + unit * UNIT
    plan = CorruptionPlan(
        flip_byte=unit_off[f] + rng.integers(0, UNIT, size=len(f)),
        flip_mask=(1 << rng.integers(0, 8, size=len(f))).astype(np.uint8),
        stale_obj=obj[s], stale_unit=unit[s],
        stale_back=rng.integers(1, 4, size=len(s)),
    )
    plan._flip_obj = obj[f]
    return plan


def stale_patches(descs: np.ndarray, plan: CorruptionPlan):
    """(byte offsets, byte values) that turn the planned units into their
    earlier-generation selves: write count (bytes 28..31) and time (40..47)."""
    o, u = plan.stale_obj, plan.stale_unit
    base = descs["body_offset"][o].astype(np.int64) + u * UNIT
    wc = (descs["write_count"][o].astype(np.int64) - plan.stale_back).astype(np.uint32)
    t = descs["unix_micro"][o] - plan.stale_back.astype(np.uint64) * np.uint64(1_000_000)
    wc_bytes = wc.view(np.uint8).reshape(-1, 4) if wc.size else np.empty((0, 4), np.uint8)
    t_bytes = np.ascontiguousarray(t).view(np.uint8).reshape(-1, 8) if t.size else np.empty((0, 8), np.uint8)
    offs = np.concatenate([(base[:, None] + np.arange(28, 32)).reshape(-1),
                           (base[:, None] + np.arange(40, 48)).reshape(-1)])
    vals = np.concatenate([wc_bytes.reshape(-1), t_bytes.reshape(-1)])
    return offs.astype(np.int64), vals.astype(np.uint8)


def apply_corruption_numpy(body: np.ndarray, descs: np.ndarray, plan: CorruptionPlan) -> None:
    offs, vals = stale_patches(descs, plan)
    body[offs] = vals
    np.bitwise_xor.at(body, plan.flip_byte, plan.flip_mask)


def apply_corruption_torch(body, descs: np.ndarray, plan: CorruptionPlan) -> None:
    """The same on a uint8 CUDA tensor."""
    import torch

    offs, vals = stale_patches(descs, plan)
    dev = body.device
    if offs.size:
        body[torch.from_numpy(offs).to(dev)] = torch.from_numpy(vals).to(dev)
    if plan.flip_byte.size:
        # a byte may be hit twice; fold the masks on the host first
        ub, inv = np.unique(plan.flip_byte, return_inverse=True)
        m = np.zeros(len(ub), dtype=np.uint8)
        np.bitwise_xor.at(m, inv, plan.flip_mask)
        idx = torch.from_numpy(ub.astype(np.int64)).to(dev)
        body[idx] = body[idx] ^ torch.from_numpy(m).to(dev)


def expected_counters(descs: np.ndarray, plan: CorruptionPlan) -> np.ndarray:
    """The 12 batch counters a validate pass must report for a batch that was
    filled and then damaged by `plan`, derived from the plan alone (no body
    needed, so it works at sizes no CPU pass can reach).

    Every planned unit is damaged exactly once (plan_corruption draws distinct
    units).  A stale unit differs in its write count (and its never-compared
    time); a bit flip is visible to the reference iff it lands in bytes [0, 40)
    of the unit (pattern.go:136-170 reads nothing else) -- a flipped name byte can
    never be trimmed away, since names are ASCII without white space inside and a
    12-byte key fills its field -- and visible to the strict count unless it lands
    in the time bytes [40, 48)."""
    from ._lib import NUM_COUNTERS

    c = np.zeros(NUM_COUNTERS, dtype=np.uint64)
    c[0] = len(descs)
    c[2] = int(descs["n_units"].astype(np.int64).sum())
    pos = (plan.flip_byte % UNIT).astype(np.int64)
    ref = pos < 40
    field = [(pos < 16), (pos >= 16) & (pos < 28), (pos >= 28) & (pos < 32), (pos >= 32) & (pos < 36),
             (pos >= 36) & (pos < 40)]
    n_stale = len(plan.stale_obj)
    c[3] = int(ref.sum()) + n_stale
    c[4], c[5] = int(field[0].sum()), int(field[1].sum())
    c[6] = int(field[2].sum()) + n_stale
    c[7], c[8] = int(field[3].sum()), int(field[4].sum())
    c[9] = int(((pos < 40) | (pos >= 48)).sum()) + n_stale
    c[1] = len(np.unique(np.concatenate([plan._flip_obj[ref], plan.stale_obj])))
    return c


class DeviceCorruption:
    """A corruption plan resident on the device, for loops that re-apply it after
    every fill (benchmarks of the cfg4 / cfg5 shape)."""

    def __init__(self, descs: np.ndarray, plan: CorruptionPlan, device):
        import torch

        offs, vals = stale_patches(descs, plan)
        self.offs = torch.from_numpy(offs).to(device)
        self.vals = torch.from_numpy(vals).to(device)
        ub, inv = np.unique(plan.flip_byte, return_inverse=True)
        m = np.zeros(len(ub), dtype=np.uint8)
        np.bitwise_xor.at(m, inv, plan.flip_mask)          # a byte may be hit twice
        self.flip_idx = torch.from_numpy(ub.astype(np.int64)).to(device)
        self.flip_mask = torch.from_numpy(m).to(device)

    def apply(self, body) -> None:
        if self.offs.numel():
            body[self.offs] = self.vals
        if self.flip_idx.numel():
            body[self.flip_idx] = body[self.flip_idx] ^ self.flip_mask


def compact_from_descs(descs: np.ndarray):
    """(segs, recs) for oval_layout_reset_compact from a descriptor table whose
    bodies lie back to back from offset 0, whose objects start at unit 0 and whose
    keys have the reference's form "ov" + 10 hex digits (object.go:48-50).  A new
    segment starts wherever bucket, worker ID, time or the key's upper 16 bits
    (process and worker index, runner.go:119-124) change."""
    from ._lib import REC_DTYPE, SEGMENT_DTYPE

    n = len(descs)
    recs = np.zeros(n, dtype=REC_DTYPE)
    if n == 0:
        return np.zeros(0, dtype=SEGMENT_DTYPE), recs
    units = descs["n_units"].astype(np.uint64)
    starts = np.concatenate([[0], np.cumsum(units)[:-1]]).astype(np.uint64) * np.uint64(UNIT)
    if (descs["body_offset"] != starts).any() or descs["first_unit"].any() or (descs["key_len"] != 12).any():
        raise ValueError("the compact form describes whole objects laid out back to back")
    k = descs["key"]
    if (k[:, 0] != ord("o")).any() or (k[:, 1] != ord("v")).any():
        raise ValueError("keys must be \"ov\" + 10 hex digits")
    digits = k[:, 2:].astype(np.int64)
    vals = np.where(digits >= ord("a"), digits - ord("a") + 10, digits - ord("0"))
    if ((vals < 0) | (vals > 15)).any() or ((digits > ord("9")) & (digits < ord("a"))).any():
        raise ValueError("keys must be \"ov\" + 10 lower-case hex digits")
    ids = np.zeros(n, dtype=np.uint64)
    for c in range(10):
        ids = (ids << np.uint64(4)) | vals[:, c].astype(np.uint64)
    base = ids & ~np.uint64(0xFFFFFF)
    recs["obj_id"] = (ids - base).astype(np.uint32)
    recs["write_count"] = descs["write_count"]
    recs["n_units"] = descs["n_units"]
    same = ((base[1:] == base[:-1]) & (descs["worker_id"][1:] == descs["worker_id"][:-1]) &
            (descs["unix_micro"][1:] == descs["unix_micro"][:-1]) & (descs["flags"][1:] == descs["flags"][:-1]) &
            (descs["bucket_len"][1:] == descs["bucket_len"][:-1]) & (descs["bucket"][1:] == descs["bucket"][:-1]).all(axis=1))
    first = np.concatenate([[0], np.flatnonzero(~same) + 1])
    segs = np.zeros(len(first), dtype=SEGMENT_DTYPE)
    segs["bucket"] = descs["bucket"][first]
    segs["bucket_len"] = descs["bucket_len"][first]
    segs["flags"] = descs["flags"][first]
    segs["worker_id"] = descs["worker_id"][first]
    segs["unix_micro"] = descs["unix_micro"][first]
    segs["key_id_base"] = base[first]
    segs["first_obj"] = first
    segs["n_obj"] = np.diff(np.concatenate([first, [n]]))
    return segs, recs

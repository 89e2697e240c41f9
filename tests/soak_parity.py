"""Long randomized comparison of the HIP path with the oracle (not collected by
pytest; run by hand on a GPU box):

    python tests/soak_parity.py [--seconds 120] [--seed 1]

Every trial draws a descriptor table (object sizes from 0 units to several
tiles, gaps from none to several tiles, parts of objects, names with white space
at an edge, worker IDs outside uint32), fills it on the device and compares every
byte with the oracle, then damages the bodies (bits, stale generations, foreign
units, zeroed runs, damage on either side of chunk and tile seams) and compares
results and counters.  Layouts are reused across trials through all rebuild
routes (in place, ahead, from compact records).  Stops at the first difference and prints the trial's seed.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import oracle  # noqa: E402
from oval_amd import engine  # noqa: E402
from oval_amd.batch import build_descs, compact_from_descs, make_keys, span_bytes  # noqa: E402

T0 = 1709558129112997


def trial(rng, ctx, layouts, k):
    n = int(rng.choice([1, 3, 17, 200, 1500, 6000]))
    hi = int(rng.choice([1, 4, 16, 17, 64, 255, 257, 1024, 5000]))
    if n * hi > 3_000_000:
        n = max(1, 3_000_000 // hi)
    units = rng.integers(0, hi + 1, size=n)
    if rng.random() < 0.3:
        units[rng.integers(0, n)] = int(rng.choice([4096, 65536 + 3, 300_000]))      # one object across many tiles
    gap = int(rng.choice([0, 0, 0, 256, 4096, 65536, 65536 * 3 + 512]))
    bucket = str(rng.choice(["test-bucket", "b", "sixteen-chars-ok!", " lead", "trail ", "a\tb"]))
    wids = rng.integers(0, 0x10000, size=n)
    if rng.random() < 0.2:
        wids = wids.astype(np.int64)
        wids[rng.integers(0, n)] = int(rng.choice([-1, 1 << 32, (1 << 40) + 5]))
    parts = rng.random() < 0.3
    descs = build_descs(bucket, make_keys(rng.integers(0, 1 << 40, size=n, dtype=np.uint64)),
                        rng.integers(0, 1 << 32, size=n), wids, T0 + rng.integers(0, 1 << 40, size=n).astype(np.uint64),
                        units * 256, first_units=(rng.integers(0, 1 << 24, size=n) if parts else 0), gap_bytes=gap)
    lay = layouts[k % len(layouts)]
    route = k % 3
    if route == 2 and gap == 0 and not parts:
        # whole objects back to back: the table can also be written on the device from compact records
        segs, recs = compact_from_descs(descs)
        lay.reset_compact(segs, recs, ahead=bool((k // 3) & 1))
        if lay.read_descs().tobytes() != descs.tobytes():
            raise AssertionError("the table written on the device differs from the host's")
    else:
        lay.reset(descs, ahead=(route == 1))
    nbytes = max(span_bytes(descs), 256) + 512
    body = torch.full((nbytes,), 0xEE, dtype=torch.uint8, device="cuda:0")
    lay.fill(body)
    ctx.sync()
    exp = np.full(nbytes, 0xEE, dtype=np.uint8)
    oracle.fill_batch(descs.view(oracle.DESC_DTYPE), exp)
    got = body.cpu().numpy()
    if not np.array_equal(got, exp):
        at = int(np.flatnonzero(got != exp)[0])
        raise AssertionError(f"fill differs at byte {at}")
    host = exp.copy()
    total = int(units.sum())
    if total:
        live = np.flatnonzero(units > 0)
        for _ in range(int(rng.integers(0, 40))):
            o = int(rng.choice(live))
            kind = int(rng.integers(0, 6))
            u = int(rng.integers(0, units[o]))
            if kind == 4:                                   # right at a chunk or tile seam of the body
                g = (int(descs["body_offset"][o]) // 256 + u)
                g = (g // 16) * 16 + int(rng.choice([-1, 0, 15, 16]))
                u = min(max(g - int(descs["body_offset"][o]) // 256, 0), int(units[o]) - 1)
            at = int(descs["body_offset"][o]) + u * 256
            if kind in (0, 4):
                host[at + int(rng.integers(0, 256))] ^= np.uint8(1 << int(rng.integers(0, 8)))
            elif kind == 1:
                host[at + 28] ^= 1                          # write count: a stale generation's only difference
            elif kind == 2:
                o2 = int(rng.choice(live))
                src = int(descs["body_offset"][o2])
                host[at:at + 256] = exp[src:src + 256]
            elif kind == 3:
                host[at:at + int(rng.integers(1, 257))] = 0
            else:                                           # a whole object stale
                a = int(descs["body_offset"][o])
                host[a + 28:a + int(units[o]) * 256:256] ^= 1
    body.copy_(torch.from_numpy(host))
    res, cnt = lay.validate(body)
    ctx.sync()
    r = engine.results_to_numpy(res, n)
    want, wcnt = oracle.validate_batch(descs.view(oracle.DESC_DTYPE), host)
    for f in want.dtype.names:
        if not np.array_equal(r[f], want[f]):
            bad = int(np.flatnonzero(r[f] != want[f])[0])
            raise AssertionError(f"{f} differs at object {bad}: got {r[f][bad]}, oracle {want[f][bad]}")
    if not np.array_equal(cnt.cpu().numpy().astype(np.uint64), wcnt):
        raise AssertionError(f"counters differ: {cnt.cpu().numpy()} vs {wcnt}")
    return total


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--small-object-bytes", type=lambda v: int(v, 0), default=None,
                    help="layouts that average fewer bytes per object take the small-object kernels "
                         "(default: the library's limits; 0xFFFFFFFFFFFFFFFF: every layout)")
    ap.add_argument("--big-object-bytes", type=lambda v: int(v, 0), default=None,
                    help="layouts that average at least this many bytes per object take the validate kernel for "
                         "multipart-sized bodies (default: the library's limit; 0: every layout)")
    args = ap.parse_args()
    if args.small_object_bytes is not None:
        engine.set_small_object_bytes(args.small_object_bytes)
    if args.big_object_bytes is not None:
        engine.set_big_object_bytes(args.big_object_bytes)
    oracle.build()
    ctx = engine.Context.on_torch_stream(0)
    seed_descs = build_descs("b", ["k"], [1], 1, T0, [256])
    layouts = [engine.Layout(ctx, seed_descs) for _ in range(2)]
    t0, k, units = time.time(), 0, 0
    last = t0
    while time.time() - t0 < args.seconds:
        seed = args.seed * 1_000_003 + k
        try:
            units += trial(np.random.default_rng(seed), ctx, layouts, k)
        except AssertionError as e:
            print(f"trial {k} (rng seed {seed}): {e}", flush=True)
            sys.exit(1)
        k += 1
        if time.time() - last > 20:
            last = time.time()
            print(f"  {k} trials, {units * 256 / 1e9:.1f} GB compared", flush=True)
    print(f"{k} trials, {units * 256 / 1e9:.1f} GB of bodies, every byte, result and counter equal to the oracle's")


if __name__ == "__main__":
    main()

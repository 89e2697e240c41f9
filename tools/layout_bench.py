"""What a batch costs when its descriptors are new every time (the benchmark's
step reuses one resident layout): oval_layout_reset per batch, alone and inside
a fill + validate step, with one layout and with two taken in turn.

    python tools/layout_bench.py [--workload cfg2] [--steps 10]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from oval_amd import engine  # noqa: E402
from oval_amd.batch import span_bytes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    batches = [bench.build_batch(args.workload, 0, 1, "workers", 0, seed)[0] for seed in (1, 2, 3)]
    nbytes = max(span_bytes(b) for b in batches)
    units = [int(b["n_units"].sum()) for b in batches]
    print(f"{args.workload}: {len(batches[0])} objects, {nbytes / 1e9:.2f} GB per batch; descriptor table "
          f"{batches[0].nbytes / 1e6:.0f} MB")
    stream = torch.cuda.Stream(0)
    torch.cuda.set_stream(stream)
    ctx = engine.Context(0, stream.cuda_stream)
    body = torch.empty(nbytes, dtype=torch.uint8, device="cuda:0")
    results = torch.empty((len(batches[0]), 4), dtype=torch.int32, device="cuda:0")
    counters = torch.zeros(engine.NUM_COUNTERS, dtype=torch.int64, device="cuda:0")

    t = time.perf_counter()
    layouts = [engine.Layout(ctx, batches[0]), engine.Layout(ctx, batches[1]), engine.Layout(ctx, batches[2])]
    ctx.sync()
    print(f"  three layouts created (allocations, pinned staging)   {1e3 * (time.perf_counter() - t):8.2f} ms")

    call, done = [], []
    for k in range(args.steps):
        t = time.perf_counter()
        layouts[0].reset(batches[k % 3])
        call.append(time.perf_counter() - t)
        ctx.sync()
        done.append(time.perf_counter() - t)
    print(f"  oval_layout_reset, new table each time: call returns after {1e3 * np.median(call):.2f} ms, tables on the "
          f"device after {1e3 * np.median(done):.2f} ms")

    from oval_amd.batch import compact_from_descs
    compact = [compact_from_descs(b) for b in batches]
    print(f"  compact form: {compact[0][1].nbytes / 1e6:.0f} MB of records + {len(compact[0][0])} segments")

    def run(n_layouts, ahead=False, use_compact=False):
        ctx.sync()
        t = time.perf_counter()
        total = 0
        for k in range(args.steps):
            lay = layouts[k % n_layouts]
            if use_compact:
                lay.reset_compact(*compact[k % 3], ahead=ahead)
            else:
                lay.reset(batches[k % 3], ahead=ahead)
            lay.fill(body)
            counters.zero_()
            lay.validate(body, results, counters)
            total += units[k % 3]
        ctx.sync()
        dt = time.perf_counter() - t
        assert int(counters.cpu()[3]) == 0
        return dt / args.steps, 512.0 * total / dt / 1e9

    for n, ahead in ((1, False), (2, False), (2, True), (3, True)):
        run(n, ahead)
        ms, gbs = run(n, ahead)
        print(f"  reset{'_ahead' if ahead else ''} + fill + validate per new batch, {n} layout(s): {1e3 * ms:8.2f} ms per batch = "
              f"{gbs:7.1f} GB/s generate+validate")
    for n, ahead in ((1, False), (2, True)):
        run(n, ahead, True)
        ms, gbs = run(n, ahead, True)
        print(f"  reset_compact{' ahead' if ahead else ''} + fill + validate per new batch, {n} layout(s): {1e3 * ms:8.2f} ms per batch = "
              f"{gbs:7.1f} GB/s generate+validate")
    call, done = [], []
    for k in range(args.steps):
        t = time.perf_counter()
        layouts[0].reset_compact(*compact[k % 3])
        call.append(time.perf_counter() - t)
        ctx.sync()
        done.append(time.perf_counter() - t)
    print(f"  oval_layout_reset_compact, new table each time: call returns after {1e3 * np.median(call):.2f} ms, tables on "
          f"the device after {1e3 * np.median(done):.2f} ms")
    lay = layouts[0]
    ctx.sync()
    t = time.perf_counter()
    for k in range(args.steps):
        lay.fill(body)
        lay.validate(body, results, counters)
    ctx.sync()
    dt = (time.perf_counter() - t) / args.steps
    print(f"  fill + validate on a resident layout:               {1e3 * dt:8.2f} ms per batch = "
          f"{512.0 * units[(args.steps - 1) % 3] / dt / 1e9:7.1f} GB/s")


if __name__ == "__main__":
    main()

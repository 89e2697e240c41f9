"""Re-validation after a blackout (README.md:200-236) as one device batch.

    python tools/blackout_bench.py [--objects N] [--workers W]

Builds the state a run leaves behind (every object exists, sizes by
DecideSize), saves and reloads the execution context in the reference's JSON
format, turns the loaded context into one descriptor table, and validates the
whole key space on the device (the bodies a GET would deliver are produced on
the device by the fill kernel here).  Prints the host-side cost of each step
next to the device pass.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from oval_amd import context as X  # noqa: E402
from oval_amd import engine  # noqa: E402

T0 = 1709558129112997


def timed(label, fn):
    t = time.perf_counter()
    out = fn()
    print(f"  {label:<44s} {1e3 * (time.perf_counter() - t):10.1f} ms", flush=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--objects", type=int, default=1_000_000)
    ap.add_argument("--workers", type=int, default=64)
    ap.add_argument("--path", default="/tmp/oval_blackout_ctx.json")
    args = ap.parse_args()
    print(f"{args.objects} objects 4k-16k, {args.workers} workers, 1 bucket")
    ec = timed("new context (Runner.init)", lambda: X.ExecutionContext.New(
        "http://localhost:9000", ["test-bucket"], args.objects, args.workers, 4096, 16384, startWorkerID=0x614))
    timed("populate (every object exists)", lambda: ec.PopulateExisting(1, 1.0))
    timed("SaveContext (JSON)", lambda: ec.SaveContext(args.path))
    print(f"  saved context: {os.path.getsize(args.path) / 1e6:.1f} MB")
    ec2 = timed("load saved context", lambda: X.ExecutionContext.LoadFile(args.path))
    descs, refs, nbytes = timed("descriptor table of existing objects", lambda: ec2.BuildExistingDescriptors(T0))
    print(f"  {len(descs)} descriptors, {nbytes / 1e9:.2f} GB of bodies")

    stream = torch.cuda.Stream(0)
    torch.cuda.set_stream(stream)
    ctx = engine.Context(0, stream.cuda_stream)
    layout = timed("layout (upload descriptors, chunk table)", lambda: (engine.Layout(ctx, descs), ctx.sync())[0])
    # the same table from 12-byte records, its descriptors written on the device
    segs, recs, nbytes_c = timed("compact table (12-byte records + segments)", lambda: ec2.BuildExistingCompact(T0))
    assert nbytes_c == nbytes and len(recs) == len(descs)
    layout_c = engine.Layout(ctx, descs[:1])
    timed("layout from the compact table, first time", lambda: (layout_c.reset_compact(segs, recs), ctx.sync()))
    timed("layout from the compact table, again", lambda: (layout_c.reset_compact(segs, recs), ctx.sync()))
    timed("layout from descriptors, again", lambda: (layout.reset(descs), ctx.sync()))
    assert layout_c.read_descs(len(descs) - 1000, 1000).tobytes() == descs[-1000:].tobytes()
    layout = layout_c
    body = torch.empty(nbytes, dtype=torch.uint8, device="cuda:0")
    results = torch.empty((len(descs), 4), dtype=torch.int32, device="cuda:0")
    counters = torch.zeros(engine.NUM_COUNTERS, dtype=torch.int64, device="cuda:0")
    layout.fill(body)
    ctx.sync()
    tv = engine.Timer(ctx)
    for _ in range(3):
        counters.zero_()
        tv.start(); layout.validate(body, results, counters); tv.stop()
        ctx.sync()
    c = counters.cpu().numpy()
    assert c[0] == len(descs) and c[1] == 0 and c[2] * 256 == nbytes
    ms = tv.elapsed_ms()
    print(f"  {'device validate of the whole key space':<44s} {ms:10.3f} ms   {nbytes / ms / 1e6:.0f} GB/s")
    os.remove(args.path)


if __name__ == "__main__":
    main()

"""Validate throughput against the fraction of bad units (device-resident cfg2 batch).

    python tools/corrupt_rates.py

Stale generations: a fraction of the objects is validated against write count 2
while holding the bytes of write count 1, so every unit of those objects is bad.
"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oval_amd import engine
from oval_amd.batch import WorkerSpace, synthetic_batch, span_bytes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--objects", type=int, default=1_000_000)
    ap.add_argument("--lib", default="", help="another build of the library (kernel experiments)")
    ap.add_argument("--fractions", default="0,1e-4,1e-3,1e-2,1e-1,1", help="stale fractions to run, comma separated")
    ap.add_argument("--no-check", action="store_true", help="do not compare the counters (timing-only experiments)")
    ap.add_argument("--no-counters", action="store_true", help="validate without a counter vector (no tallies are kept)")
    args = ap.parse_args()
    if args.lib:
        from oval_amd import _lib
        _lib.LIB_PATH = os.path.abspath(args.lib)
    stream = torch.cuda.Stream(0)
    torch.cuda.set_stream(stream)
    ctx = engine.Context(0, stream.cuda_stream)
    n_obj = args.objects - args.objects % 64
    sp = WorkerSpace(num_obj=n_obj, num_worker=64)
    descs = synthetic_batch(sp, range(64), "bench-bucket", 4096, 16384, seed=1, unix_micro=1_700_000_000_000_000)
    n = len(descs)
    units = int(descs["n_units"].sum())
    body = torch.empty(span_bytes(descs), dtype=torch.uint8, device="cuda:0")
    lay = engine.Layout(ctx, descs)
    lay.fill(body)
    results = torch.empty((n, 4), dtype=torch.int32, device="cuda:0")
    counters = torch.zeros(engine.NUM_COUNTERS, dtype=torch.int64, device="cuda:0")
    rng = np.random.default_rng(7)
    for frac in [float(x) for x in args.fractions.split(",")]:
        d2 = descs.copy()
        stale = rng.random(n) < frac
        d2["write_count"][stale] += 1
        lay2 = engine.Layout(ctx, d2)
        t = engine.Timer(ctx)
        ms = []
        for _ in range(5):
            counters.zero_()
            t.start()
            if args.no_counters:
                from oval_amd import _lib
                _lib.check(_lib.lib().oval_validate(ctx.handle, lay2._h, body.data_ptr(), body.numel(), results.data_ptr(), None))
            else:
                lay2.validate(body, results, counters)
            t.stop()
            torch.cuda.synchronize()
            ms.append(t.elapsed_ms())
        c = counters.cpu().numpy()
        bad_units = int(descs["n_units"][stale].sum())
        if not args.no_check and not args.no_counters:
            assert int(c[3]) == bad_units and int(c[1]) == int(stale.sum()), (c, bad_units, stale.sum())
        best = min(ms[1:])
        print(f"stale objects {frac:7.4f}  bad units {bad_units:9d}  validate {best:8.3f} ms  "
              f"{256.0 * units / best / 1e6:8.1f} GB/s", flush=True)


if __name__ == "__main__":
    main()

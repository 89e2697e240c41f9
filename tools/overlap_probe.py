"""Does a fill of one batch run well next to the validate of another?  Two
contexts (streams), two body buffers of the cfg2 shape; per round one fill and
one validate, first one after the other on one stream, then side by side.

    python tools/overlap_probe.py [--rounds 20]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

import bench  # noqa: E402
from oval_amd import engine  # noqa: E402
from oval_amd.batch import span_bytes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--rounds", type=int, default=20)
    args = ap.parse_args()
    descs = bench.build_batch(args.workload, 0, 1, "workers", 0, 1)[0]
    nbytes, units = span_bytes(descs), int(descs["n_units"].sum())
    s1, s2 = torch.cuda.Stream(0), torch.cuda.Stream(0)
    c1, c2 = engine.Context(0, s1.cuda_stream), engine.Context(0, s2.cuda_stream)
    l1, l2 = engine.Layout(c1, descs), engine.Layout(c2, descs)
    bodies = [torch.empty(nbytes, dtype=torch.uint8, device="cuda:0") for _ in range(2)]
    res = [torch.empty((len(descs), 4), dtype=torch.int32, device="cuda:0") for _ in range(2)]
    cnt = [torch.zeros(engine.NUM_COUNTERS, dtype=torch.int64, device="cuda:0") for _ in range(2)]
    l1.fill(bodies[0]); l1.fill(bodies[1]); c1.sync()

    def sequential():
        for k in range(args.rounds):
            l1.fill(bodies[k & 1])
            l1.validate(bodies[(k & 1) ^ 1], res[0], cnt[0])
        c1.sync()

    def side_by_side():
        for k in range(args.rounds):
            # the validate of round k reads what the fill of round k-1 wrote: order the two streams
            e1, e2 = torch.cuda.Event(), torch.cuda.Event()
            l1.fill(bodies[k & 1])
            l2.validate(bodies[(k & 1) ^ 1], res[1], cnt[1])
            e1.record(s1); e2.record(s2)
            s2.wait_event(e1); s1.wait_event(e2)
        c1.sync(); c2.sync()

    for name, fn in (("one after the other", sequential), ("side by side", side_by_side)):
        fn()
        t = time.perf_counter()
        fn()
        dt = (time.perf_counter() - t) / args.rounds
        print(f"  fill + validate {name:<20s} {1e3 * dt:7.3f} ms per round = {512.0 * units / dt / 1e9:7.1f} GB/s")
    assert int(cnt[0][3]) == 0 and int(cnt[1][3]) == 0


if __name__ == "__main__":
    main()

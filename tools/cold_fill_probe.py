"""A fill whose lookup tables are NOT in the memory-side cache (the first fill of a
batch): before every timed fill a 2 GiB buffer is written, which evicts them.
    python tools/cold_fill_probe.py <objects> <min>-<max>
Prints the mean fill time of 8 cold fills and of 8 fills right after a validate pass."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oval_amd import engine  # noqa: E402
from oval_amd.batch import WorkerSpace, span_bytes, synthetic_batch  # noqa: E402

n, (lo, hi) = int(sys.argv[1]), (int(x) for x in sys.argv[2].split("-"))
n -= n % 25
descs = synthetic_batch(WorkerSpace(num_obj=n, num_worker=25, start_worker_id=0x561B), range(25), "test-bucket", lo, hi,
                        seed=1, unix_micro=1709558129112997)
ctx = engine.Context.on_torch_stream(0)
layout = engine.Layout(ctx, descs)
body = torch.empty(span_bytes(descs), dtype=torch.uint8, device="cuda:0")
evict = torch.empty(2 << 30, dtype=torch.uint8, device="cuda:0")
units = int(descs["n_units"].sum())


def run(cold):
    ms = []
    for _ in range(8):
        if cold:
            evict.fill_(1)
        else:
            layout.validate(body)
        t = engine.Timer(ctx)
        t.start(); layout.fill(body); t.stop()
        ctx.sync()
        ms.append(t.elapsed_ms())
    return float(np.mean(ms))


layout.fill(body); ctx.sync()
c, w = run(True), run(False)
print(f"n {n} sizes {lo}-{hi} nowarm={os.environ.get('OVAL_EXP_NOWARM', '0')}: cold fill {c:.4f} ms {256 * units / c / 1e6:.1f} GB/s, "
      f"after a validate {w:.4f} ms {256 * units / w / 1e6:.1f} GB/s", flush=True)

"""Does memory handed back to the driver and allocated again serve a stream as well
as memory allocated in a fresh process?  cfg3 (64 x 256 MiB) after nothing, after
9.5 GB were allocated, written and given back with hipFree (torch.cuda.empty_cache),
after a cfg2 measurement, and with the freed block kept in torch's cache:

    python tools/placement_probe.py fresh | after_alloc | after_cfg2 | keep_cache
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

mode = sys.argv[1]
args = argparse.Namespace(gpus=1, backend="nccl", collective="torch", fill_threads=0, validate_threads=0, sharding="workers",
                          seed=1, batch_gib=64.0, objects=0)
R = bench.Rank(args)
if mode in ("after_alloc", "keep_cache"):
    x = torch.empty(9_520_000_000, dtype=torch.uint8, device="cuda:0")
    x.fill_(1)
    torch.cuda.synchronize()
    del x
    if mode == "after_alloc":
        torch.cuda.empty_cache()
elif mode == "after_cfg2":
    bench.measure(R, "cfg2", 5, 1)
    torch.cuda.empty_cache()
m = bench.measure(R, "cfg3", 10, 2)
print(mode, "cfg3 fill %.1f validate %.1f GB/s" % (m["fill_gbs"], m["val_gbs"]), flush=True)
R.close()

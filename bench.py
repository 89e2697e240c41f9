#!/usr/bin/env python3
"""Headline benchmark: device-resident pattern generate+validate GB/s.

One step = one fill pass + one validate pass over one resident batch of
synthetic objects (default: 1,000,000 objects of 4k-16k per GPU, sizes by
DecideSize, pinned unix time), followed for N > 1 by the single all-reduce of
the counter vector.  Objects shard by worker key range: rank r owns the key
space of process ID r (64 workers), so no body byte crosses GPUs and scaling
is weak.  Prints ONE JSON line on rank 0.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

T0 = 1709558129112997          # README.md:105, the pinned unix time (microseconds)
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak
WORKLOADS = {
    # name: (objects per GPU, workers, min size, max size, description)
    "cfg2": (1_000_000, 64, 4096, 16384, "1M objects 4k-16k (DecideSize), fill+validate batch"),
    "cfg3": (64, 64, 256 << 20, 256 << 20, "64 objects x 256 MiB, fill+validate"),
    "cfg5": (1_250_000, 50, 4096, 1 << 20, "1.25M objects 4k-1M per GPU (10M over 8), resident sub-batches"),
}


def build_batch(name, rank, n_obj_override, seed, size_override=None):
    from oval_amd.batch import WorkerSpace, synthetic_batch

    n_obj, n_worker, lo, hi, _ = WORKLOADS[name]
    if n_obj_override:
        n_obj = n_obj_override
    if size_override:
        lo, hi = size_override
    n_worker = min(n_worker, n_obj)
    n_obj -= n_obj % n_worker
    space = WorkerSpace(num_obj=n_obj, num_worker=n_worker, runner_id=rank, start_worker_id=0x561B)
    return synthetic_batch(space, range(n_worker), "test-bucket", lo, hi, seed=seed, unix_micro=T0)


def cpu_baseline(descs, budget_s, threads):
    """The oracle (a C port of the reference's pattern.go) on this host's cores,
    over a bounded prefix of the same batch: fill then validate, per slice."""
    from oracle import oracle as O
    from oval_amd.batch import span_bytes

    O.build()
    D = descs.view(O.DESC_DTYPE)
    slice_objs = max(1, min(len(descs), 20000))

    def run_slice(lo, out):
        d = D[lo:lo + slice_objs].copy()
        d["body_offset"] -= d["body_offset"][0]
        body = np.empty(span_bytes(d), dtype=np.uint8)
        t = time.perf_counter()
        O.fill_batch(d, body)
        res, _ = O.validate_batch(d, body)
        dt = time.perf_counter() - t
        assert (res["first_bad_unit"] == O.NO_BAD).all()
        out.append((int(d["n_units"].sum()), dt))

    done, t_start, lo = [], time.perf_counter(), 0
    while lo < len(descs) and time.perf_counter() - t_start < budget_s:
        ths, outs = [], []
        for _ in range(threads):
            if lo >= len(descs):
                break
            o = []
            th = threading.Thread(target=run_slice, args=(lo, o))
            th.start()
            ths.append(th); outs.append(o)
            lo += slice_objs
        for th in ths:
            th.join()
        done += [o[0] for o in outs]
    wall = time.perf_counter() - t_start
    units = sum(u for u, _ in done)
    return {
        "value": round(512.0 * units / wall / 1e9, 4), "unit": "GB/s", "cores": threads, "kind": "port",
        "sample": f"oracle fill+validate of the first {min(lo, len(descs))} objects ({units} units, "
                  f"{units * 256 / 1e9:.2f} GB) of the same batch, {wall:.1f} s, {threads} thread(s)",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--objects", type=int, default=0, help="objects per GPU (default: the workload's)")
    ap.add_argument("--fill-threads", type=int, default=0, help="workgroup size of the fill kernel (0 = default)")
    ap.add_argument("--validate-threads", type=int, default=0, help="workgroup size of the validate kernel")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline (0 = skip)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--collective", default="torch", choices=["torch", "capi"],
                    help="who runs the counter all-reduce: torch.distributed (default) or the library's own "
                         "RCCL entry point oval_allreduce_counters (needs one GPU per rank)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo to "
                    "rehearse more ranks than GPUs: ranks then share devices round-robin)")
    ap.add_argument("--sizes", default="", help="override object sizes, e.g. 65536-65536 (bytes, multiples of 256)")
    ap.add_argument("--batch-gib", type=float, default=64.0,
                    help="largest body buffer; a bigger workload runs as consecutive sub-batches through it")
    args = ap.parse_args()

    # stdout carries the one JSON line and nothing else: libraries that print
    # banners there (RCCL does on communicator creation) go to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch

    from oval_amd import engine
    from oval_amd.batch import span_bytes, split_batches
    from oval_amd.dist import all_reduce, all_reduce_counters

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the pattern engine has no CPU path")
    if args.backend != "nccl":
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    stream = torch.cuda.Stream(local)
    torch.cuda.set_stream(stream)
    ctx = engine.Context(local, stream.cuda_stream)
    ctx.set_tuning(args.fill_threads, args.validate_threads)

    comm = None
    if args.collective == "capi":
        uid = [engine.Comm.unique_id() if rank == 0 else None]
        if dist:
            dist.broadcast_object_list(uid, src=0)
        comm = engine.Comm(ctx, world, rank, uid[0])

    size_override = tuple(int(x) for x in args.sizes.split("-")) if args.sizes else None
    descs = build_batch(args.workload, rank, args.objects, args.seed, size_override)
    n_obj, units = len(descs), int(descs["n_units"].sum())
    # One resident layout per sub-batch (descriptors + chunk table stay on the
    # device), one body buffer reused by all of them.
    parts = split_batches(descs, int(args.batch_gib * (1 << 30)))
    nbytes = sum(span_bytes(p) for p in parts)
    layouts = [engine.Layout(ctx, p) for p in parts]
    body = torch.empty(max(span_bytes(p) for p in parts), dtype=torch.uint8, device=f"cuda:{local}")
    results = [torch.empty((len(p), 4), dtype=torch.int32, device=body.device) for p in parts]
    counters = torch.zeros(engine.NUM_COUNTERS, dtype=torch.int64, device=body.device)

    def step(timers=None):
        counters.zero_()
        for k, layout in enumerate(layouts):
            tf, tv = timers[k] if timers else (None, None)
            if tf: tf.start()
            layout.fill(body)
            if tf: tf.stop()
            if tv: tv.start()
            layout.validate(body, results[k], counters)
            if tv: tv.stop()
        if comm:                               # the one collective of a batch
            comm.all_reduce_counters(counters)
        else:
            all_reduce_counters(counters)

    for _ in range(args.warmup):
        step()
    timers = [[(engine.Timer(ctx), engine.Timer(ctx)) for _ in layouts] for _ in range(args.steps)]
    torch.cuda.synchronize()
    if dist: dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for per_step in timers:
        step(per_step)
    torch.cuda.synchronize()
    if dist: dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0

    # per step: the sum over sub-batches (one launch each); then the mean over steps
    fill_ms = float(np.mean([sum(tf.elapsed_ms() for tf, _ in per_step) for per_step in timers]))
    val_ms = float(np.mean([sum(tv.elapsed_ms() for _, tv in per_step) for per_step in timers]))
    tot = counters.cpu().numpy()
    stats = torch.tensor([wall, float(units)], dtype=torch.float64, device=body.device)
    if dist:
        mx = all_reduce(stats.clone(), dist.ReduceOp.MAX)
        sm = all_reduce(stats.clone(), dist.ReduceOp.SUM)
        wall, units_all = float(mx[0]), float(sm[1])
    else:
        units_all = float(units)
    # the batch was clean: nothing may be reported bad, everything must have been checked
    assert int(tot[1]) == 0 and int(tot[3]) == 0, f"clean batch reported corrupt: {tot}"
    assert int(tot[2]) == int(units_all), (int(tot[2]), units_all)

    if rank == 0:
        ms_per_step = wall / args.steps * 1e3
        value = 512.0 * units_all / (wall / args.steps) / 1e9
        fill_gbs = 256.0 * units / (fill_ms / 1e3) / 1e9
        val_gbs = 256.0 * units / (val_ms / 1e3) / 1e9
        # kernel names as rocprofv3 reports them
        # the library gives objects that average under 256 KiB the one-level lookup
        one_level = "true" if nbytes / max(n_obj, 1) < (256 << 10) else "false"
        fill_name = "fill_kernel<%d, %s>" % (args.fill_threads or 256, one_level)
        val_name = "validate_kernel<%d>" % (args.validate_threads or 256)
        n_fill = sum(l.launches()[0] for l in layouts)
        n_val = sum(l.launches()[1] for l in layouts)
        dom, dom_ms, dom_gbs, dom_n = ((val_name, val_ms, val_gbs, n_val) if val_ms >= fill_ms
                                       else (fill_name, fill_ms, fill_gbs, n_fill))
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                if not (args.objects or args.sizes):             # measured on the default shape only
                    # the file holds HBM bytes per pass over the batch (PMC passes); per launch here
                    per_pass = json.load(open(tpath)).get(args.workload, {}).get(dom)
                    traffic = per_pass // max(dom_n, 1) if per_pass else None
            except Exception:
                traffic = None
        out = {
            "metric": "device-resident pattern generate+validate GB/s",
            "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {WORKLOADS[args.workload][4]}", "objects_per_gpu": n_obj,
                       "units_per_gpu": units, "bytes_per_gpu": nbytes, "sub_batches": len(parts), "sharding": "worker key range (process ID = rank)", "collective": args.collective,
                       "fill_threads": args.fill_threads or 256, "validate_threads": args.validate_threads or 256},
            "frac_of_hbm_peak": round(value / world / HBM_PEAK_GBS, 4),
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(dom_gbs, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(dom_gbs / HBM_PEAK_GBS, 4), "traffic": traffic,
                         # a pass over the batch is `launches_per_step` equal launches of this kernel; the
                         # HIP events bracket the pass, small helper kernels and launch gaps included
                         "launches_per_step": dom_n, "algorithmic_bytes_per_launch": 256 * units // max(dom_n, 1),
                         "avg_launch_ms": round(dom_ms / max(dom_n, 1), 4)},
            "kernels": {"fill_ms": round(fill_ms, 4), "fill_GBps": round(fill_gbs, 1),
                        "validate_ms": round(val_ms, 4), "validate_GBps": round(val_gbs, 1),
                        "fill_launches": n_fill, "validate_launches": n_val},
        }
        if world == 1 and args.cpu_seconds > 0:
            # all of this process's share of the host cores, and one core beside it
            cores = max(1, min(16, len(os.sched_getaffinity(0))))
            out["cpu_baseline"] = cpu_baseline(descs, args.cpu_seconds / 2, cores)
            one = cpu_baseline(descs, args.cpu_seconds / 2, 1)
            out["cpu_baseline"]["single_thread_value"] = one["value"]
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if comm:
        torch.cuda.synchronize()
        comm.close()
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

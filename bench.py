#!/usr/bin/env python3
"""Headline benchmark: device-resident pattern generate+validate GB/s.

One step = one fill pass + one validate pass over one resident batch of
synthetic objects (default: 1,000,000 objects of 4k-16k per GPU, sizes by
DecideSize, pinned unix time), followed for N > 1 by the single all-reduce of
the counter vector.  Objects shard by worker key range (--sharding workers, the
default): ONE key space, process ID 0, of N x the per-GPU workers and objects
(internal/runner/runner.go:117-124), cut into contiguous worker ranges, one per
rank; --sharding process gives rank r the whole key space of process ID r
instead (the reference's multi-process mode).  Either way no body byte crosses
GPUs and scaling is weak.  Prints ONE JSON line on rank 0.

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


def build_batch(name, rank, world, sharding, n_obj_override, seed, size_override=None):
    """This rank's descriptors and a description of how the key space was cut."""
    from oval_amd.batch import WorkerSpace, shard_workers, synthetic_batch

    n_obj, n_worker, lo, hi, _ = WORKLOADS[name]
    if n_obj_override:
        n_obj = n_obj_override
    if size_override:
        lo, hi = size_override
    if sharding == "process":
        n_worker = min(n_worker, n_obj)
        n_obj -= n_obj % n_worker
        space = WorkerSpace(num_obj=n_obj, num_worker=n_worker, runner_id=rank, start_worker_id=0x561B)
        mine = range(n_worker)
        how = f"process ID = rank, {n_worker} workers x {space.per_worker} objects each"
    else:
        # one key space for the whole job: per-GPU workers x N workers (num_worker < 256,
        # cmd/root.go:170-173), per-GPU objects x N objects, rank r owns a contiguous worker range
        per_gpu_workers = max(1, min(n_worker, 255 // world, n_obj))
        n_obj -= n_obj % per_gpu_workers
        space = WorkerSpace(num_obj=n_obj * world, num_worker=per_gpu_workers * world, runner_id=0,
                            start_worker_id=0x561B)
        mine = shard_workers(space.num_worker, world, rank)
        how = (f"one key space (process ID 0): {space.num_worker} workers x {space.per_worker} objects = "
               f"{space.num_obj} objects, contiguous worker ranges per rank (rank 0: workers {mine.start}-{mine.stop - 1})")
    return synthetic_batch(space, mine, "test-bucket", lo, hi, seed=seed, unix_micro=T0), how


def host_cpu_share():
    """Threads a CPU-bound pool may use here: the CPUs this process may run on,
    cut to the container's declared share where there is one (a cgroup CPU quota,
    or the pool size the environment sets for OpenMP)."""
    info = {"os_cpu_count": os.cpu_count(), "affinity_cpus": len(os.sched_getaffinity(0))}
    share = info["affinity_cpus"]
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            info["cgroup_cpu_quota"] = round(int(quota) / int(period), 2)
            share = min(share, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    omp = os.environ.get("OMP_NUM_THREADS", "")
    if omp.isdigit() and int(omp) > 0:
        info["omp_num_threads"] = int(omp)
        share = min(share, int(omp))
    info["threads_used"] = max(1, share)
    return info


def go_toolchain():
    """`go version` if a Go toolchain is on this box (the reference is Go)."""
    import shutil
    import subprocess

    exe = shutil.which("go")
    if not exe:
        return "absent (no `go` on PATH): the reference's own pattern package cannot be timed here"
    try:
        return subprocess.run([exe, "version"], capture_output=True, text=True, timeout=20).stdout.strip()
    except Exception as e:  # noqa: BLE001
        return f"present but `go version` failed: {e}"


def _cpu_run(descs, budget_s, threads, variant):
    """One timing of the oracle over a bounded prefix of the batch, `threads`
    object streams side by side (one goroutine per worker in the reference,
    internal/runner/runner.go:186-214).  Returns (GB/s, objects, units, seconds)."""
    from oracle import oracle as O
    from oval_amd.batch import span_bytes

    D = descs.view(O.DESC_DTYPE)
    slice_objs = max(1, min(len(descs), 20000 if variant == "port" else 5000))

    def run_slice(lo, out):
        d = D[lo:lo + slice_objs].copy()
        d["body_offset"] -= d["body_offset"][0]
        if variant == "port":
            body = np.empty(span_bytes(d), dtype=np.uint8)
            O.fill_batch(d, body)
            res, _ = O.validate_batch(d, body)
            assert (res["first_bad_unit"] == O.NO_BAD).all()
            out.append(int(d["n_units"].sum()))
        else:
            units, bad = O.refshape_generate_valid(d)
            assert bad == 0
            out.append(units)

    # the batch is walked slice by slice, again from its start when it has been
    # used up, until the time budget is spent: a fast host gets through cfg2 in
    # well under a second, too short a sample to time on its own
    done, t_start, lo, reached = [], time.perf_counter(), 0, 0
    while time.perf_counter() - t_start < budget_s:
        ths, outs = [], []
        for _ in range(threads):
            if lo >= len(descs):
                lo = 0
            o = []
            th = threading.Thread(target=run_slice, args=(lo, o))
            th.start()
            ths.append(th); outs.append(o)
            lo += slice_objs
            reached = max(reached, min(lo, len(descs)))
        for th in ths:
            th.join()
        done += [o[0] for o in outs]
    wall = time.perf_counter() - t_start
    units = sum(done)
    return 512.0 * units / wall / 1e9, reached, units, wall


def cpu_baseline(descs, budget_s):
    """The oracle (a C restatement of the reference's pattern.go) on this host's
    cores over a bounded prefix of the same batch, generate then validate, in two
    variants: the plain port (template copy per unit) and one shaped like the
    reference's code (per unit: formatted print, small allocations, a clock read,
    append copies; pattern.go:95-115), each on one thread and on the host's share
    of cores.  `value` is the faster one (the port) on all of them."""
    from oracle import oracle as O

    O.build()
    share = host_cpu_share()
    n = share["threads_used"]
    port_all = _cpu_run(descs, budget_s / 4, n, "port")
    port_one = _cpu_run(descs, budget_s / 4, 1, "port")
    ref_all = _cpu_run(descs, budget_s / 4, n, "refshape")
    ref_one = _cpu_run(descs, budget_s / 4, 1, "refshape")
    return {
        "value": round(port_all[0], 4), "unit": "GB/s", "cores": n, "kind": "port",
        "sample": f"oracle generate+validate over the first {port_all[1]} objects of the same batch, repeated until the "
                  f"time budget was spent: {port_all[2]} units ({port_all[2] * 256 / 1e9:.2f} GB) in {port_all[3]:.1f} s, "
                  f"{n} thread(s)",
        "single_thread_value": round(port_one[0], 4),
        "reference_shaped": {
            "value": round(ref_all[0], 4), "single_thread_value": round(ref_one[0], 4), "cores": n,
            "sample": f"first {ref_all[1]} objects, repeated: {ref_all[2]} units in {ref_all[3]:.1f} s; per unit: printf-formatted "
                      "names, 4 small allocations, clock_gettime, 52 four-byte appends, 256-byte append copy, "
                      "256-byte read + 2 string copies + TrimSpace compare (pattern.go:95-115,124-170)",
        },
        "host": share,
        "go_toolchain": go_toolchain(),
    }


def measure_traffic(argv_tail, kernel_prefix, is_fill, launches_per_pass):
    """HBM bytes per launch of the dominant kernel from PMC counters, measured NOW
    on this box: a child run of this same bench under `rocprofv3 --pmc` (one
    counter, its own run; the profiler cannot be attached from inside).  Counter
    units are KiB; FETCH_SIZE tallies the 128-byte requests of a wide streaming
    read at 64 bytes each on gfx950, so read bytes = 2 x FETCH_SIZE x 1024.
    Returns (bytes per launch, how) or (None, why not)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not found"
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        return None, "this run is itself under a profiler"
    counter = "WRITE_SIZE" if is_fill else "FETCH_SIZE"
    passes = 3
    out = tempfile.mkdtemp(prefix="oval_pmc_")
    try:
        cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", out, "--", sys.executable,
               os.path.join(ROOT, "bench.py"), "--steps", str(passes - 1), "--warmup", "1", "--cpu-seconds", "0",
               "--no-traffic"] + argv_tail
        p = subprocess.run(cmd, cwd=out, capture_output=True, text=True, timeout=240,
                           env=dict(os.environ, TMPDIR="/tmp"))
        if p.returncode != 0:
            return None, f"rocprofv3 --pmc {counter} child failed (rc {p.returncode})"
        total, n = 0.0, 0
        for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == counter and ("oval::" + kernel_prefix) in r["Kernel_Name"]:
                    total += float(r["Counter_Value"])
                    n += 1
        if n == 0:
            return None, f"no {counter} rows for {kernel_prefix}"
        per_pass = total * 1024.0 * (1.0 if is_fill else 2.0) / passes
        return int(per_pass / max(launches_per_pass, 1)), (
            f"measured in this run: child `rocprofv3 --pmc {counter}` of the same command, {n} launches over "
            f"{passes} passes" + ("" if is_fill else ", x2 gfx950 FETCH_SIZE correction"))
    except Exception as e:  # noqa: BLE001
        return None, f"PMC child run failed: {e}"
    finally:
        shutil.rmtree(out, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--objects", type=int, default=0, help="objects per GPU (default: the workload's)")
    ap.add_argument("--fill-threads", type=int, default=0, help="workgroup size of the fill kernel (0 = default)")
    ap.add_argument("--validate-threads", type=int, default=0, help="workgroup size of the validate kernel")
    ap.add_argument("--cpu-seconds", type=float, default=16.0, help="budget of the CPU baseline (0 = skip)")
    ap.add_argument("--corrupt", type=float, default=0.0,
                    help="fraction of units damaged after every fill (half bit flips, half stale generations; "
                         "seeded per rank): the cfg4 / cfg5 shape.  The injection is inside the step, so `value` "
                         "is not the headline figure then")
    ap.add_argument("--verify-oracle", action="store_true",
                    help="rank 0 recomputes every rank's shard with the CPU oracle and requires the all-reduced "
                         "counters to equal the oracle's over the whole batch (small --objects only)")
    ap.add_argument("--global-table", action="store_true",
                    help="also exchange the job-wide first-bad-unit table (4 bytes per object of the whole job, "
                         "minimum over ranks) after every validate: the optional second form of the collective")
    ap.add_argument("--verify-plan", action="store_true",
                    help="require the all-reduced counters to equal what the damage plans of all ranks imply "
                         "(no CPU pass over the bodies: works at full size)")
    ap.add_argument("--no-traffic", action="store_true",
                    help="do not measure roofline.traffic with a child rocprofv3 --pmc run (N=1 only; ~15 s)")
    ap.add_argument("--sharding", default="workers", choices=["workers", "process"],
                    help="workers: one key space cut into worker ranges (default); process: rank r = process ID r")
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
    from oval_amd.batch import DeviceCorruption, expected_counters, plan_corruption, span_bytes, split_batches
    from oval_amd.dist import all_reduce, all_reduce_counters, all_reduce_first_bad

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the pattern engine has no CPU path")
    if args.backend != "nccl":
        local %= torch.cuda.device_count()
    elif world > torch.cuda.device_count():
        raise SystemExit(f"--gpus {world} over RCCL needs {world} visible GPUs, this node shows "
                         f"{torch.cuda.device_count()}: one rank per GPU (use --backend gloo to rehearse more "
                         "ranks than GPUs)")
    if args.collective == "capi" and world > torch.cuda.device_count():
        raise SystemExit(f"--collective capi runs RCCL in the library and needs one GPU per rank: {world} ranks, "
                         f"{torch.cuda.device_count()} visible GPUs")
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
    descs, sharding_how = build_batch(args.workload, rank, world, args.sharding, args.objects, args.seed, size_override)
    n_obj, units = len(descs), int(descs["n_units"].sum())
    # One resident layout per sub-batch (descriptors + chunk table stay on the
    # device), one body buffer reused by all of them.
    parts = split_batches(descs, int(args.batch_gib * (1 << 30)))
    nbytes = sum(span_bytes(p) for p in parts)
    layouts = [engine.Layout(ctx, p) for p in parts]
    body = torch.empty(max(span_bytes(p) for p in parts), dtype=torch.uint8, device=f"cuda:{local}")
    results = [torch.empty((len(p), 4), dtype=torch.int32, device=body.device) for p in parts]
    counters = torch.zeros(engine.NUM_COUNTERS, dtype=torch.int64, device=body.device)
    damage = None
    planned = np.zeros(engine.NUM_COUNTERS, dtype=np.uint64)
    planned[0], planned[2] = n_obj, units
    if args.corrupt > 0:
        # sub-batch k of rank r is damaged with seed (seed, r, k): reproducible on any other rank
        plans = [plan_corruption(p, args.corrupt, seed=args.seed * 1000003 + rank * 1009 + k) for k, p in enumerate(parts)]
        damage = [DeviceCorruption(p, pl, body.device) for p, pl in zip(parts, plans)]
        planned = sum(expected_counters(p, pl) for p, pl in zip(parts, plans))

    table = my_table = None
    if args.global_table:
        # where this rank's objects sit in the table of the whole job: ranks in order
        counts = [n_obj]
        if dist:
            counts = [None] * world
            dist.all_gather_object(counts, n_obj)
        base = sum(counts[:rank])
        table = torch.empty(sum(counts), dtype=torch.int32, device=body.device)
        offs = np.concatenate([[0], np.cumsum([len(p) for p in parts])])
        my_table = [table[base + int(offs[k]): base + int(offs[k + 1])] for k in range(len(parts))]

    def step(timers=None):
        counters.zero_()
        if table is not None:
            table.fill_(-1)                    # 0xFFFFFFFF everywhere: no bad unit known
        for k, layout in enumerate(layouts):
            tf, tv = timers[k] if timers else (None, None)
            if tf: tf.start()
            layout.fill(body)
            if tf: tf.stop()
            if damage: damage[k].apply(body)
            if tv: tv.start()
            layout.validate(body, results[k], counters)
            if tv: tv.stop()
            if table is not None:
                engine.gather_first_bad(ctx, results[k], len(parts[k]), my_table[k])
        if comm:                               # the one collective of a batch
            comm.all_reduce_counters(counters)
            if table is not None and world > 1:
                comm.all_reduce_first_bad(table)
        else:
            all_reduce_counters(counters)
            if table is not None:
                all_reduce_first_bad(table)

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
    # everything must have been checked; a clean batch may report nothing bad
    assert int(tot[2]) == int(units_all), (int(tot[2]), units_all)
    if not damage:
        assert int(tot[1]) == 0 and int(tot[3]) == 0, f"clean batch reported corrupt: {tot}"
    plan_ok = None
    if args.verify_plan:
        # what every rank's plan implies, gathered as plain lists (not through the all-reduce under test)
        mine = [int(x) for x in planned]
        if dist:
            every = [None] * world
            dist.all_gather_object(every, mine)
        else:
            every = [mine]
        want = np.sum(np.array(every, dtype=np.uint64), axis=0)
        assert np.array_equal(tot.astype(np.uint64), want), f"counters {tot} differ from the plans' {want}"
        plan_ok = True
    verified = None
    if args.verify_oracle and rank == 0:
        # the whole job on the CPU: every rank's shard, its damage, the oracle's counters summed
        from oracle import oracle as O
        from oval_amd.batch import apply_corruption_numpy

        expect = np.zeros(engine.NUM_COUNTERS, dtype=np.uint64)
        for r in range(world):
            d_r, _ = build_batch(args.workload, r, world, args.sharding, args.objects, args.seed, size_override)
            for k, p in enumerate(split_batches(d_r, int(args.batch_gib * (1 << 30)))):
                host = np.zeros(span_bytes(p), dtype=np.uint8)
                O.fill_batch(p.view(O.DESC_DTYPE), host)
                if args.corrupt > 0:
                    apply_corruption_numpy(host, p, plan_corruption(p, args.corrupt, seed=args.seed * 1000003 + r * 1009 + k))
                expect += O.validate_batch(p.view(O.DESC_DTYPE), host)[1]
        assert np.array_equal(tot.astype(np.uint64), expect), f"counters {tot} differ from the oracle's {expect}"
        verified = [int(x) for x in expect]
        if table is not None:
            # the whole job's first bad units, rank after rank, from the oracle
            want = []
            for r in range(world):
                d_r, _ = build_batch(args.workload, r, world, args.sharding, args.objects, args.seed, size_override)
                for k, p in enumerate(split_batches(d_r, int(args.batch_gib * (1 << 30)))):
                    host = np.zeros(span_bytes(p), dtype=np.uint8)
                    O.fill_batch(p.view(O.DESC_DTYPE), host)
                    if args.corrupt > 0:
                        apply_corruption_numpy(host, p, plan_corruption(p, args.corrupt, seed=args.seed * 1000003 + r * 1009 + k))
                    want.append(O.validate_batch(p.view(O.DESC_DTYPE), host)[0]["first_bad_unit"])
            want = np.concatenate(want)
            got = table.cpu().numpy().view(np.uint32)
            assert np.array_equal(got, want), "the job-wide first-bad-unit table differs from the oracle's"

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
        traffic, traffic_source = None, "none: no stored PMC pass for this shape"
        if world == 1 and not args.no_traffic and not damage:
            tail = ["--workload", args.workload, "--seed", str(args.seed), "--batch-gib", str(args.batch_gib),
                    "--sharding", args.sharding]
            if args.objects: tail += ["--objects", str(args.objects)]
            if args.sizes: tail += ["--sizes", args.sizes]
            if args.fill_threads: tail += ["--fill-threads", str(args.fill_threads)]
            if args.validate_threads: tail += ["--validate-threads", str(args.validate_threads)]
            is_fill = dom.startswith("fill")
            traffic, traffic_source = measure_traffic(tail, "fill_kernel" if is_fill else "validate_kernel", is_fill, dom_n)
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                if traffic is None and not (args.objects or args.sizes):   # stored for the default shapes only
                    # the file holds HBM bytes per pass over the batch (PMC passes); per launch here
                    stored = json.load(open(tpath))
                    per_pass = stored.get(args.workload, {}).get(dom)
                    traffic = per_pass // max(dom_n, 1) if per_pass else None
                    if traffic:
                        traffic_source = (traffic_source + "; fell back to profiles/traffic.json: " + stored.get("source", "an earlier rocprofv3 --pmc pass")
                                          + "; NOT measured in this run")
            except Exception:
                traffic = None
        out = {
            "metric": "device-resident pattern generate+validate GB/s",
            "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {WORKLOADS[args.workload][4]}", "objects_per_gpu": n_obj,
                       "units_per_gpu": units, "bytes_per_gpu": nbytes, "sub_batches": len(parts), "sharding": sharding_how, "collective": args.collective,
                       "fill_threads": args.fill_threads or 256, "validate_threads": args.validate_threads or 256},
            "frac_of_hbm_peak": round(value / world / HBM_PEAK_GBS, 4),
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(dom_gbs, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(dom_gbs / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_source": traffic_source,
                         # a pass over the batch is `launches_per_step` equal launches of this kernel; the
                         # HIP events bracket the pass, small helper kernels and launch gaps included
                         "launches_per_step": dom_n, "algorithmic_bytes_per_launch": 256 * units // max(dom_n, 1),
                         "avg_launch_ms": round(dom_ms / max(dom_n, 1), 4)},
            "kernels": {"fill_ms": round(fill_ms, 4), "fill_GBps": round(fill_gbs, 1),
                        "validate_ms": round(val_ms, 4), "validate_GBps": round(val_gbs, 1),
                        "fill_launches": n_fill, "validate_launches": n_val},
        }
        if damage:
            out["config"]["corrupt_fraction"] = args.corrupt
            out["counters"] = [int(x) for x in tot]
        if verified is not None:
            out["counters_equal_oracle"] = True
        if plan_ok:
            out["counters_equal_plan"] = True
        if table is not None:
            out["config"]["global_table_objects"] = int(table.numel())
            if verified is not None:
                out["global_table_equals_oracle"] = True
            out["global_table_bad_objects"] = int((table != -1).sum().item())
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(descs, args.cpu_seconds)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if comm:
        torch.cuda.synchronize()
        comm.close()
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

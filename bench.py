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
GPUs and scaling is weak: every rank holds the same number of workers and
objects whatever N is.  The collective is the library's own RCCL entry point
(oval_allreduce_counters -> ncclAllReduce) under the RCCL backend.  After the
headline workload (the default cfg2) the same invocation times cfg3, cfg4 and a
batch of the reference's smallest objects and reports them under `also`.  Prints ONE JSON line on rank 0.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python bench.py --gpus N --steps K --warmup W        (starts its own N rank processes)
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
    # name: (objects per GPU, workers per GPU, min size, max size, description).  The reference allows
    # fewer than 256 workers in one key space (cmd/root.go:170-173), so with at most 25 per GPU every
    # N up to 10 gets the same per-GPU shape (8 GPUs: 200 workers; cfg5: 200 x 50,000 = 10M objects).
    "cfg2": (1_000_000, 25, 4096, 16384, "1M objects 4k-16k (DecideSize), fill+validate batch"),
    "cfg3": (64, 16, 256 << 20, 256 << 20, "64 objects x 256 MiB, fill+validate"),
    "cfg5": (1_250_000, 25, 4096, 1 << 20, "1.25M objects 4k-1M per GPU (10M over 8), resident sub-batches"),
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
        same = "the same per-GPU shape at every N <= %d" % (255 // max(1, min(n_worker, n_obj)))
        how = (f"one key space (process ID 0): {space.num_worker} workers x {space.per_worker} objects = "
               f"{space.num_obj} objects, contiguous worker ranges per rank ({per_gpu_workers} workers x "
               f"{space.per_worker} objects per GPU, {same}; rank 0: workers {mine.start}-{mine.stop - 1})")
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
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                info["cpu_model"] = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
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


def measure_traffic(argv_tail, launches):
    """HBM bytes per launch of the two kernels from PMC counters, measured NOW on
    this box: two child runs of this same bench under `rocprofv3 --pmc`, one
    counter each (FETCH_SIZE and WRITE_SIZE do not fit one pass; the profiler
    cannot be attached from inside).  Counter units are KiB; FETCH_SIZE tallies the
    128-byte requests of a wide streaming read (16 bytes per lane) at 64 bytes
    each on gfx950, so read bytes = 2 x FETCH_SIZE x 1024 for the validate kernel
    and the table pre-pass.  The fill kernel reads nothing but 64-byte records with
    scalar loads, which the counter tallies at full size (calibrated with
    tools/scalar_fetch_probe.hip: 1.000 x a table read once that way, 0.500 x with
    wide loads), so its read bytes = FETCH_SIZE x 1024.  The validate kernel's own
    descriptor reads (scalar, about 1 % of its bytes) are doubled with the rest: an
    upper bound.  `launches` = {kernel name prefix: launches per pass}.  Returns ({prefix: {"read": B, "write": B}} per
    launch, how) or (None, why not)."""
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
    passes = 3
    got = {k: {} for k in launches}
    for counter, key, scale in (("FETCH_SIZE", "read", 2.0), ("WRITE_SIZE", "write", 1.0)):
        out = tempfile.mkdtemp(prefix="oval_pmc_")
        try:
            cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", out, "--", sys.executable,
                   os.path.join(ROOT, "bench.py"), "--steps", str(passes - 1), "--warmup", "1", "--cpu-seconds", "0",
                   "--no-traffic", "--no-also", "--collective", "torch"] + argv_tail
            p = subprocess.run(cmd, cwd=out, capture_output=True, text=True, timeout=240,
                               env=dict(os.environ, TMPDIR="/tmp"))
            if p.returncode != 0:
                return None, f"rocprofv3 --pmc {counter} child failed (rc {p.returncode})"
            seen = 0
            for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
                for r in csv.DictReader(open(f)):
                    if r["Counter_Name"] != counter:
                        continue
                    for prefix in launches:
                        if ("oval::" + prefix) in r["Kernel_Name"]:
                            # (the small-object fill reads its descriptors with vector loads that run over
                            # neighbouring records: wide requests, doubled like the others — at counter value they
                            # would come to half the descriptor table the kernel must read)
                            k_scale = 1.0 if (key == "read" and prefix == "fill_kernel") else scale
                            got[prefix][key] = got[prefix].get(key, 0.0) + float(r["Counter_Value"]) * 1024.0 * k_scale
                            seen += 1
            if seen == 0:
                return None, f"no {counter} rows for the pattern kernels"
        except Exception as e:  # noqa: BLE001
            return None, f"PMC child run failed: {e}"
        finally:
            shutil.rmtree(out, ignore_errors=True)
    per_launch = {k: {rw: int(v / passes / max(launches[k], 1)) for rw, v in d.items()} for k, d in got.items()}
    return per_launch, (f"measured in this run: two child `rocprofv3 --pmc` runs of the same command (FETCH_SIZE, "
                        f"WRITE_SIZE), {passes} passes each; x2 gfx950 FETCH_SIZE correction for the wide reads of validate_kernel "
                        f"and warm_tables_kernel, x1 for the 64-byte scalar reads of fill_kernel")


def self_launch(n, timeout_s):
    """`python bench.py --gpus N` without a launcher: N fresh rank processes, started
    before this process has imported torch or touched a GPU; rank 0's one JSON
    line is relayed, any rank's failure ends the others and this process with
    its code.  Nothing is retried."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OVAL_BENCH_LAUNCHER="self")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    deadline = time.monotonic() + timeout_s
    rc = 0
    while True:
        codes = [p.poll() for p in procs]
        bad = [c for c in codes if c not in (None, 0)]
        if bad:
            rc = bad[0]
            break
        if all(c == 0 for c in codes):
            break
        if time.monotonic() > deadline:
            print(f"bench.py: ranks still running after {timeout_s} s", file=sys.stderr)
            rc = 124
            break
        time.sleep(0.1)
    if rc:
        for p in procs:                       # the exact processes started above, nothing else
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
        print(f"bench.py: a rank failed (exit code {rc}); see its messages above", file=sys.stderr)
        return rc if 0 < rc < 256 else 1
    sys.stdout.write(procs[0].stdout.read().decode())
    sys.stdout.flush()
    return 0


def rehearse_launch(args, json_fd):
    """The launch path without the engine: rendezvous over gloo, this rank's shard of
    the key space, one sum over ranks on host memory, one line from rank 0 that
    carries no measurement."""
    import torch
    import torch.distributed as dist

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    descs, how = build_batch(args.workload, rank, world, args.sharding, args.objects, args.seed)
    t = torch.tensor([len(descs), int(descs["n_units"].sum()), 1], dtype=torch.int64)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dist.all_reduce(t)
    if rank == 0:
        out = {"metric": "device-resident pattern generate+validate GB/s", "value": None, "unit": "GB/s",
               "n_gpus": world, "rehearsal": "launch only: nothing was filled, validated or timed",
               "config": {"workload": f"{args.workload}: {WORKLOADS[args.workload][4]}", "objects_per_gpu": len(descs),
                          "objects_all": int(t[0]), "units_all": int(t[1]), "ranks_seen": int(t[2]), "sharding": how,
                          "launcher": os.environ.get("OVAL_BENCH_LAUNCHER", "external" if world > 1 else "none")}}
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def compat_path(seconds=2.0):
    """The path the reference itself calls, one object per call (DecideSize ->
    Generate -> Valid, 4-16 KiB objects; internal/runner/worker.go:77-109), timed by
    build/compat_bench (tools/compat_bench.cpp) as a child process: 255 blocking
    threads, and 14 threads x 32 workers with calls in two steps.  Host-bound
    figures, reported beside the device-resident headline, never as `value`."""
    import re
    import subprocess

    exe = os.path.join(ROOT, "build", "compat_bench")
    if not os.path.exists(exe):
        return {"skipped": "build/compat_bench is missing (python -c 'import __graft_entry__ as g; g.build()')"}
    out = {}
    for name, argv, env in (("blocking_255_threads", [str(seconds), "255", "2", "0", "1"], {}),
                            ("two_step_14_threads_x_32_workers", [str(seconds), "14", "2", "0", "1"], {"COMPAT_ASYNC": "32"})):
        try:
            p = subprocess.run([exe] + argv, capture_output=True, text=True, timeout=120, env=dict(os.environ, **env))
            m = re.search(r"([0-9]+) put-cycles/s\s+([0-9.]+) GB/s generate\+validate\s+errors ([0-9]+)", p.stdout)
            if p.returncode != 0 or not m:
                out[name] = {"failed": (p.stderr or p.stdout)[-300:]}
                continue
            sysm = re.search(r"per put-cycle ([0-9.]+) us user \+ ([0-9.]+) us system", p.stdout)
            out[name] = {"put_cycles_per_s": int(m.group(1)), "GBps": float(m.group(2)), "errors": int(m.group(3)),
                         "user_us_per_put_cycle": float(sysm.group(1)) if sysm else None,
                         "system_us_per_put_cycle": float(sysm.group(2)) if sysm else None}
        except Exception as e:  # noqa: BLE001
            out[name] = {"failed": str(e)}
    out["what"] = ("one object per call through the host mirror of internal/pattern (DecideSize, Generate, Valid; 4-16 KiB), "
                   f"{seconds:g} s each; GB/s = bytes generated + bytes validated; host-bound, not the headline")
    return out


class Rank:
    """What one rank process holds for the whole run."""

    def __init__(self, args):
        import torch

        from oval_amd import engine

        self.args = args
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={self.world}")
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the pattern engine has no CPU path")
        n_dev = torch.cuda.device_count()
        if args.backend != "nccl":
            local %= n_dev
        elif self.world > n_dev:
            raise SystemExit(f"--gpus {self.world} over RCCL needs {self.world} visible GPUs, this node shows {n_dev}: "
                             "one rank per GPU (use --backend gloo to rehearse more ranks than GPUs)")
        collective = args.collective
        self.collective_note = None
        if collective == "auto":
            # the collective of the path is ncclAllReduce from the library's own entry point; under gloo
            # (ranks sharing GPUs) RCCL cannot run, torch.distributed carries the sum instead
            collective = "capi" if args.backend == "nccl" else "torch"
        if collective == "capi" and self.world > n_dev:
            raise SystemExit(f"--collective capi runs RCCL in the library and needs one GPU per rank: {self.world} ranks, "
                             f"{n_dev} visible GPUs")
        self.local = local
        torch.cuda.set_device(local)
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if args.backend == "nccl":
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=torch.device("cuda", local))
            else:
                dist.init_process_group(args.backend, rank=self.rank, world_size=self.world)
            self.dist = dist
        self.stream = torch.cuda.Stream(local)
        torch.cuda.set_stream(self.stream)
        self.ctx = engine.Context(local, self.stream.cuda_stream)
        self.ctx.set_tuning(args.fill_threads, args.validate_threads)
        self.device = torch.device("cuda", local)
        self.comm = None
        self.ranks_seen = 1
        if collective == "capi":
            why = None
            try:
                uid = [engine.Comm.unique_id() if self.rank == 0 else None]
                if self.dist:
                    self.dist.broadcast_object_list(uid, src=0)
                self.comm = engine.Comm(self.ctx, self.world, self.rank, uid[0])
                self.ranks_seen = self.comm.info()[0]        # from the communicator, not from argv
            except Exception as e:  # noqa: BLE001
                if args.collective == "capi":
                    raise                                    # asked for by name: no substitute
                why = str(e)
            if args.collective == "auto":
                # Every rank takes the same road: if the library's communicator could not be made on any of
                # them, all carry the sum through torch.distributed (RCCL as well) and the line says so.
                bad = torch.tensor([1 if why else 0], dtype=torch.int32, device=torch.device("cuda", local))
                if self.dist:
                    self.dist.all_reduce(bad, op=self.dist.ReduceOp.MAX)
                if int(bad.item()):
                    if self.comm:
                        self.comm.close()
                        self.comm = None
                    collective = "torch"
                    self.collective_note = "capi unavailable, torch.distributed carries the sum: " + (why or "failed on another rank")
        if self.comm is None:
            self.ranks_seen = self.dist.get_world_size() if self.dist else 1
        self.collective = collective

    def close(self):
        import torch

        if self.comm:
            torch.cuda.synchronize()
            self.comm.close()
        if self.dist:
            self.dist.barrier()
            self.dist.destroy_process_group()


def measure(R, workload, steps, warmup, corrupt=0.0, objects=0, sizes=None, verify_plan=False, verify_oracle=False,
            global_table=False):
    """`warmup` untimed steps, then exactly `steps` timed ones of `workload` on every
    rank, bracketed by a barrier and a device synchronisation on both sides; the
    slowest rank's time counts.  Returns what the JSON line is made of."""
    import torch

    from oval_amd import engine
    from oval_amd.batch import DeviceCorruption, expected_counters, plan_corruption, span_bytes, split_batches
    from oval_amd.dist import all_reduce, all_reduce_counters, all_reduce_first_bad

    args, dist, ctx, comm, rank, world = R.args, R.dist, R.ctx, R.comm, R.rank, R.world
    descs, sharding_how = build_batch(workload, rank, world, args.sharding, objects, args.seed, sizes)
    n_obj, units = len(descs), int(descs["n_units"].sum())
    # One resident layout per sub-batch (descriptors + chunk table stay on the
    # device), one body buffer reused by all of them.
    parts = split_batches(descs, int(args.batch_gib * (1 << 30)))
    nbytes = sum(span_bytes(p) for p in parts)
    layouts = [engine.Layout(ctx, p) for p in parts]
    body = torch.empty(max(span_bytes(p) for p in parts), dtype=torch.uint8, device=R.device)
    results = [torch.empty((len(p), 4), dtype=torch.int32, device=body.device) for p in parts]
    counters = torch.zeros(engine.NUM_COUNTERS, dtype=torch.int64, device=body.device)
    damage = None
    planned = np.zeros(engine.NUM_COUNTERS, dtype=np.uint64)
    planned[0], planned[2] = n_obj, units
    if corrupt > 0:
        # sub-batch k of rank r is damaged with seed (seed, r, k): reproducible on any other rank
        plans = [plan_corruption(p, corrupt, seed=args.seed * 1000003 + rank * 1009 + k) for k, p in enumerate(parts)]
        damage = [DeviceCorruption(p, pl, body.device) for p, pl in zip(parts, plans)]
        planned = sum(expected_counters(p, pl) for p, pl in zip(parts, plans))

    table = my_table = None
    if global_table:
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

    for _ in range(warmup):
        step()
    timers = [[(engine.Timer(ctx), engine.Timer(ctx)) for _ in layouts] for _ in range(steps)]
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
    if verify_plan:
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
    verified = table_ok = None
    if verify_oracle and rank == 0:
        # the whole job on the CPU: every rank's shard, its damage, the oracle's counters summed
        from oracle import oracle as O
        from oval_amd.batch import apply_corruption_numpy

        expect = np.zeros(engine.NUM_COUNTERS, dtype=np.uint64)
        want_table = []
        for r in range(world):
            d_r, _ = build_batch(workload, r, world, args.sharding, objects, args.seed, sizes)
            for k, p in enumerate(split_batches(d_r, int(args.batch_gib * (1 << 30)))):
                host = np.zeros(span_bytes(p), dtype=np.uint8)
                O.fill_batch(p.view(O.DESC_DTYPE), host)
                if corrupt > 0:
                    apply_corruption_numpy(host, p, plan_corruption(p, corrupt, seed=args.seed * 1000003 + r * 1009 + k))
                res_k, cnt_k = O.validate_batch(p.view(O.DESC_DTYPE), host)
                expect += cnt_k
                want_table.append(res_k["first_bad_unit"])
        assert np.array_equal(tot.astype(np.uint64), expect), f"counters {tot} differ from the oracle's {expect}"
        verified = [int(x) for x in expect]
        if table is not None:
            got = table.cpu().numpy().view(np.uint32)
            assert np.array_equal(got, np.concatenate(want_table)), "the job-wide first-bad-unit table differs from the oracle's"
            table_ok = True

    ms_per_step = wall / steps * 1e3
    out = {
        "workload": workload, "descs": descs, "sharding_how": sharding_how, "n_obj": n_obj, "units": units,
        "units_all": units_all, "nbytes": nbytes, "sub_batches": len(parts), "ms_per_step": ms_per_step,
        "value": 512.0 * units_all / (wall / steps) / 1e9, "fill_ms": fill_ms, "val_ms": val_ms,
        "fill_gbs": 256.0 * units / (fill_ms / 1e3) / 1e9, "val_gbs": 256.0 * units / (val_ms / 1e3) / 1e9,
        "n_fill": sum(l.launches()[0] for l in layouts), "n_val": sum(l.launches()[1] for l in layouts),
        "counters": [int(x) for x in tot], "damaged": bool(damage), "plan_ok": plan_ok, "verified": verified,
        "table_ok": table_ok, "table_objects": int(table.numel()) if table is not None else None,
        "table_bad": int((table != -1).sum().item()) if table is not None else None,
    }
    for l in layouts:
        l.close()
    # The buffers go back to torch's caching allocator, NOT to the driver: a body buffer allocated after 9.5 GB
    # were handed back with hipFree gets memory that serves a read stream 6 % slower (cfg3 validate 6.97 against
    # 7.39 TB/s in a fresh process or with the freed block kept cached; tools/placement_probe.py).
    del body, results, counters, damage, table, my_table, timers
    return out


def kernel_roofline(name, gbs, ms, launches, units, traffic):
    """One kernel's roofline object: algorithmic bytes (256 per unit) over the mean
    duration of a pass, measured with HIP events on the launching stream."""
    r = {"bound": "hbm", "kernel": name, "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None,
         # a pass over the batch is `launches_per_step` equal launches of this kernel; the HIP events
         # bracket the pass, small helper kernels (table pre-pass, result initialisation, tally fold)
         # and launch gaps included
         "launches_per_step": launches, "algorithmic_bytes_per_launch": 256 * units // max(launches, 1),
         "pass_ms": round(ms, 4), "avg_launch_ms": round(ms / max(launches, 1), 4)}
    if traffic:
        r["traffic_read"], r["traffic_write"] = traffic.get("read", 0), traffic.get("write", 0)
        r["traffic"] = r["traffic_read"] + r["traffic_write"]
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--objects", type=int, default=0, help="objects per GPU (default: the workload's)")
    ap.add_argument("--fill-threads", type=int, default=0, help="workgroup size of the fill kernel (0 = default)")
    ap.add_argument("--validate-threads", type=int, default=0, help="workgroup size of the validate kernel")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="budget of the CPU baseline (0 = skip)")
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
                    help="do not measure roofline traffic with child rocprofv3 --pmc runs (N=1 only; ~20 s)")
    ap.add_argument("--no-also", action="store_true",
                    help="do not run cfg3 and cfg4 after the headline workload (they are run, and reported under "
                         "`also`, when the headline is the default cfg2 without damage)")
    ap.add_argument("--no-compat", action="store_true",
                    help="do not time the one-object-per-call path (build/compat_bench, two child runs of 2 s) under `also`")
    ap.add_argument("--also-steps", type=int, default=10, help="timed steps of the cfg3 run under `also` (cfg4: half)")
    ap.add_argument("--small-object-bytes", default="",
                    help="F or F,V: layouts that average fewer than F / V bytes per object take the small-object fill / "
                         "validate kernel (default: the library's limits; 0: never)")
    ap.add_argument("--big-object-bytes", type=int, default=-1,
                    help="layouts that average at least this many bytes per object take the validate kernel for "
                         "multipart-sized bodies (default: the library's limit; 0: every layout)")
    ap.add_argument("--sharding", default="workers", choices=["workers", "process"],
                    help="workers: one key space cut into worker ranges (default); process: rank r = process ID r")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--collective", default="auto", choices=["auto", "torch", "capi"],
                    help="who runs the counter all-reduce: the library's own RCCL entry point "
                         "oval_allreduce_counters (capi; needs one GPU per rank) or torch.distributed (torch); "
                         "auto = capi under the RCCL backend, torch under gloo")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo to "
                    "rehearse more ranks than GPUs: ranks then share devices round-robin)")
    ap.add_argument("--sizes", default="", help="override object sizes, e.g. 65536-65536 (bytes, multiples of 256)")
    ap.add_argument("--batch-gib", type=float, default=64.0,
                    help="largest body buffer; a bigger workload runs as consecutive sub-batches through it")
    ap.add_argument("--launch-timeout", type=float, default=1500.0,
                    help="seconds after which a self-launched job (--gpus N without a launcher) is given up")
    ap.add_argument("--rehearse-launch", action="store_true",
                    help="start the ranks, rendezvous (gloo), cut the key space, sum the ranks' unit counts and "
                         "print a line WITHOUT a measurement (value null): the launch path alone, runnable on a "
                         "box without a GPU")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around this process: it becomes one, before torch or the GPU is touched
        raise SystemExit(self_launch(args.gpus, args.launch_timeout))

    # stdout carries the one JSON line and nothing else: libraries that print
    # banners there (RCCL does on communicator creation) go to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    if args.rehearse_launch:
        rehearse_launch(args, json_fd)
        return

    R = Rank(args)
    rank, world = R.rank, R.world
    if args.small_object_bytes:
        from oval_amd import engine

        engine.set_small_object_bytes(*(int(x) for x in args.small_object_bytes.split(",")))
    if args.big_object_bytes >= 0:
        from oval_amd import engine

        engine.set_big_object_bytes(args.big_object_bytes)
    size_override = tuple(int(x) for x in args.sizes.split("-")) if args.sizes else None
    m = measure(R, args.workload, args.steps, args.warmup, corrupt=args.corrupt, objects=args.objects,
                sizes=size_override, verify_plan=args.verify_plan, verify_oracle=args.verify_oracle,
                global_table=args.global_table)

    also = None
    if not args.no_also and args.workload == "cfg2" and args.corrupt == 0 and not size_override:
        # the other two single-GPU configurations under the same clock: cfg3 (streaming roofline) and
        # cfg4 (cfg2 with 1e-5 of its units damaged after every fill, counters held to the damage plan)
        c3 = measure(R, "cfg3", max(1, args.also_steps), 2)
        c4 = measure(R, "cfg2", max(1, args.also_steps // 2), 1, corrupt=1e-5, objects=args.objects, verify_plan=True)
        also = {
            "cfg3": {"workload": WORKLOADS["cfg3"][4], "value": round(c3["value"], 2), "unit": "GB/s",
                     "steps": max(1, args.also_steps), "ms_per_step": round(c3["ms_per_step"], 4),
                     "bytes_per_gpu": c3["nbytes"], "fill_ms": round(c3["fill_ms"], 4),
                     "validate_ms": round(c3["val_ms"], 4), "fill_GBps": round(c3["fill_gbs"], 1),
                     "validate_GBps": round(c3["val_gbs"], 1),
                     "frac": round(c3["value"] / world / HBM_PEAK_GBS, 4)},
            "cfg4": {"workload": "cfg2 with 1e-5 of its units damaged after every fill (half bit flips, half stale "
                                 "generations); the injection is inside the step",
                     "value": round(c4["value"], 2), "unit": "GB/s", "steps": max(1, args.also_steps // 2),
                     "ms_per_step": round(c4["ms_per_step"], 4), "validate_ms": round(c4["val_ms"], 4),
                     "validate_GBps": round(c4["val_gbs"], 1), "counters": c4["counters"],
                     "counters_equal_plan": bool(c4["plan_ok"])},
        }
        # the reference's smallest objects (one unit and up, pattern.go:23-34): the kernels that resolve all
        # objects of a chunk at once, under the same clock (four objects for every one of the headline batch)
        sm = measure(R, "cfg2", max(1, args.also_steps), 2, objects=4 * (args.objects or 1_000_000), sizes=(256, 1024))
        also["small_objects"] = {
            "workload": f"{sm['n_obj']} objects of 256-1024 bytes (DecideSize), fill+validate", "value": round(sm["value"], 2),
            "unit": "GB/s", "steps": max(1, args.also_steps), "ms_per_step": round(sm["ms_per_step"], 4),
            "bytes_per_gpu": sm["nbytes"], "descriptor_bytes_per_gpu": 64 * sm["n_obj"],
            "fill_ms": round(sm["fill_ms"], 4), "validate_ms": round(sm["val_ms"], 4),
            "fill_GBps": round(sm["fill_gbs"], 1), "validate_GBps": round(sm["val_gbs"], 1),
            "frac": round(sm["value"] / world / HBM_PEAK_GBS, 4)}
        if world == 1 and rank == 0 and not args.no_compat:
            also["compat_path"] = compat_path()

    if rank == 0:
        units, n_obj, nbytes = m["units"], m["n_obj"], m["nbytes"]
        # kernel names as rocprofv3 reports them
        # the library gives objects that average under 256 KiB the one-level lookup
        one_level = "true" if nbytes / max(n_obj, 1) < (256 << 10) else "false"
        fill_name = "fill_kernel<%d, %s>" % (args.fill_threads or 256, one_level)
        val_name = "validate_kernel<%d>" % (args.validate_threads or 256)
        # objects of a few units take the kernels that resolve a chunk's objects at once (the library picks by the
        # average object size of every layout; the limits are its defaults unless --small-object-bytes set them)
        from oval_amd import engine as _engine
        lim = [int(x) for x in args.small_object_bytes.split(",")] if args.small_object_bytes else list(_engine.SMALL_OBJECT_BYTES_DEFAULT)
        lim = lim * 2 if len(lim) == 1 else lim
        if m["sub_batches"] == 1 and nbytes // max(n_obj, 1) < lim[0]:
            fill_name = "fill_small_kernel<%d>" % (args.fill_threads or 256)
        if m["sub_batches"] == 1 and nbytes // max(n_obj, 1) < lim[1]:
            val_name = "validate_small_kernel<%d>" % (args.validate_threads or 256)
        big = args.big_object_bytes if args.big_object_bytes >= 0 else _engine.BIG_OBJECT_BYTES_DEFAULT
        if m["sub_batches"] == 1 and nbytes // max(n_obj, 1) >= big:
            val_name = "validate_big_kernel"
        traffic, traffic_source = None, "none: not measured"
        if world == 1 and not args.no_traffic and not m["damaged"]:
            tail = ["--workload", args.workload, "--seed", str(args.seed), "--batch-gib", str(args.batch_gib),
                    "--sharding", args.sharding]
            if args.objects: tail += ["--objects", str(args.objects)]
            if args.sizes: tail += ["--sizes", args.sizes]
            if args.fill_threads: tail += ["--fill-threads", str(args.fill_threads)]
            if args.validate_threads: tail += ["--validate-threads", str(args.validate_threads)]
            if args.small_object_bytes: tail += ["--small-object-bytes", args.small_object_bytes]
            if args.big_object_bytes >= 0: tail += ["--big-object-bytes", str(args.big_object_bytes)]
            fk, vk = fill_name.split("<")[0], val_name.split("<")[0]
            traffic, traffic_source = measure_traffic(tail, {fk: m["n_fill"], vk: m["n_val"],
                                                             "warm_tables_kernel": m["n_fill"]})
        rf = kernel_roofline(fill_name, m["fill_gbs"], m["fill_ms"], m["n_fill"], units,
                             traffic and traffic.get(fill_name.split("<")[0]))
        rv = kernel_roofline(val_name, m["val_gbs"], m["val_ms"], m["n_val"], units,
                             traffic and traffic.get(val_name.split("<")[0]))
        if traffic and traffic.get("warm_tables_kernel"):
            rf["table_prepass_read_per_launch"] = traffic["warm_tables_kernel"].get("read", 0)
        dom = dict(rv if m["val_ms"] >= m["fill_ms"] else rf)
        dom["traffic_source"] = traffic_source
        out = {
            "metric": "device-resident pattern generate+validate GB/s",
            "value": round(m["value"], 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(m["ms_per_step"], 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {WORKLOADS[args.workload][4]}", "objects_per_gpu": n_obj,
                       "units_per_gpu": units, "bytes_per_gpu": nbytes, "sub_batches": m["sub_batches"],
                       "sharding": m["sharding_how"], "collective": R.collective, "ranks_seen": R.ranks_seen,
                       "backend": args.backend if world > 1 else "none (one rank)",
                       "launcher": os.environ.get("OVAL_BENCH_LAUNCHER", "external" if world > 1 else "none"),
                       "fill_threads": args.fill_threads or 256, "validate_threads": args.validate_threads or 256},
            "frac_of_hbm_peak": round(m["value"] / world / HBM_PEAK_GBS, 4),
            "roofline": dom, "roofline_fill": rf, "roofline_validate": rv,
            "kernels": {"fill_ms": round(m["fill_ms"], 4), "fill_GBps": round(m["fill_gbs"], 1),
                        "validate_ms": round(m["val_ms"], 4), "validate_GBps": round(m["val_gbs"], 1),
                        "fill_launches": m["n_fill"], "validate_launches": m["n_val"]},
        }
        if R.collective_note:
            out["config"]["collective_note"] = R.collective_note
        if m["damaged"]:
            out["config"]["corrupt_fraction"] = args.corrupt
            out["counters"] = m["counters"]
        if m["verified"] is not None:
            out["counters_equal_oracle"] = True
        if m["plan_ok"]:
            out["counters_equal_plan"] = True
        if m["table_objects"] is not None:
            out["config"]["global_table_objects"] = m["table_objects"]
            if m["table_ok"]:
                out["global_table_equals_oracle"] = True
            out["global_table_bad_objects"] = m["table_bad"]
        if also:
            out["also"] = also
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(m["descs"], args.cpu_seconds)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    R.close()


if __name__ == "__main__":
    main()

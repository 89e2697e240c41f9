"""bench.py's contract on a real GPU: exactly one JSON line on stdout with the
fields the driver reads, for one rank and for a two-rank rehearsal (gloo, both
ranks on the one GPU of a test box; the 8-GPU runs use RCCL with one GPU each)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline"}


def _one_line(cmd):
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_single_rank_line():
    d = _one_line([sys.executable, "bench.py", "--objects", "40000", "--steps", "3", "--warmup", "1",
                   "--cpu-seconds", "1"])
    assert REQUIRED <= set(d) and "cpu_baseline" in d
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["unit"] == "GB/s" and d["dtype"] == "u8" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["value"] > 100 and 0 < d["roofline"]["frac"] < 1 and d["roofline"]["bound"] == "hbm"
    assert d["roofline"]["peak"] == 8000.0 and d["roofline"]["launches_per_step"] >= 1
    assert set(d["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"} and d["cpu_baseline"]["kind"] == "port"
    assert "workload" in d["config"] and "model" not in d["config"]
    cb = d["cpu_baseline"]
    assert cb["reference_shaped"]["value"] > 0 and cb["host"]["threads_used"] == cb["cores"] >= 1
    assert "go_toolchain" in cb and cb["host"]["os_cpu_count"] >= cb["host"]["threads_used"]
    # HBM traffic of the dominant kernel, measured by a child PMC run of the same command: within a few
    # per cent of the algorithmic bytes (descriptor and table reads on top of the bodies)
    r = d["roofline"]
    rf, rv = d["roofline_fill"], d["roofline_validate"]
    assert rf["kernel"].startswith("fill_kernel") and rv["kernel"].startswith("validate_kernel")
    assert r["kernel"] in (rf["kernel"], rv["kernel"]) and r["frac"] == min(rf["frac"], rv["frac"])
    for k in (rf, rv):
        assert k["launches_per_step"] >= 1 and k["avg_launch_ms"] > 0 and 0 < k["frac"] < 1
        assert abs(k["avg_launch_ms"] * k["launches_per_step"] - k["pass_ms"]) < 0.01 * k["pass_ms"] + 1e-3
    if r["traffic_source"].startswith("measured in this run"):
        # both kernels, two PMC passes: the fill writes its bytes once and reads tables on top, the
        # validate reads its bytes once (plus descriptors) and writes next to nothing
        assert 1.0 <= rv["traffic"] / rv["algorithmic_bytes_per_launch"] < 1.15
        assert 0.99 <= rf["traffic_write"] / rf["algorithmic_bytes_per_launch"] < 1.02
        assert rf["traffic_read"] < 0.25 * rf["algorithmic_bytes_per_launch"]
        assert r["traffic"] in (rf["traffic"], rv["traffic"])
    else:
        # no profiler on this box (or this test itself runs under one): the line must say why, and must
        # not pass a number off as measured
        assert r["traffic"] is None and len(r["traffic_source"]) > 8, r["traffic_source"]
    # the other two single-GPU configurations under the same clock
    a = d["also"]
    assert set(a) == {"cfg3", "cfg4", "small_objects", "compat_path"}
    so = a["small_objects"]
    assert so["value"] > 100 and so["fill_ms"] > 0 and so["validate_ms"] > 0 and 0 < so["frac"] < 1
    assert so["descriptor_bytes_per_gpu"] == 64 * 160000 and 256 * 160000 <= so["bytes_per_gpu"] <= 1024 * 160000
    cp = a["compat_path"]
    if "skipped" not in cp:
        for k in ("blocking_255_threads", "two_step_14_threads_x_32_workers"):
            assert cp[k]["errors"] == 0 and cp[k]["put_cycles_per_s"] > 50_000, cp[k]
        # calls in two steps never sleep in the library: next to no system time per put-cycle
        assert cp["two_step_14_threads_x_32_workers"]["system_us_per_put_cycle"] < 2.0
    assert a["cfg3"]["bytes_per_gpu"] == 64 * (256 << 20) and a["cfg3"]["value"] > 1000 and 0 < a["cfg3"]["frac"] < 1
    assert a["cfg3"]["fill_ms"] > 0 and a["cfg3"]["validate_ms"] > 0
    assert a["cfg4"]["counters_equal_plan"] is True and a["cfg4"]["counters"][3] > 0 and a["cfg4"]["value"] > 100
    assert d["config"]["ranks_seen"] == 1 and d["config"]["collective"] in ("capi", "torch")


def test_two_rank_rehearsal_sums_units_over_ranks():
    # no launcher: bench.py starts its own two rank processes
    d = _one_line([sys.executable, "bench.py", "--gpus", "2", "--backend", "gloo", "--objects", "40000",
                   "--steps", "2", "--warmup", "1", "--cpu-seconds", "0", "--no-also"])
    assert d["roofline"]["traffic"] is None                 # PMC passes are taken at N = 1 only
    assert REQUIRED <= set(d) and d["n_gpus"] == 2 and "cpu_baseline" not in d
    assert d["config"]["launcher"] == "self" and d["config"]["ranks_seen"] == 2
    assert d["config"]["collective"] == "torch"             # gloo: two ranks on one GPU cannot run RCCL
    # whole-job value: both ranks' units over the slowest rank's time
    per_gpu = d["config"]["units_per_gpu"]
    assert abs(d["value"] - 512.0 * 2 * per_gpu / (d["ms_per_step"] / 1e3) / 1e9) / d["value"] < 0.05


@pytest.mark.parametrize("sharding", ["workers", "process"])
def test_two_rank_rehearsal_with_damage_counters_equal_the_oracle(sharding):
    """cfg5's shape in small: two ranks, each damages its own shard (seeded per
    rank), one all-reduce of the 12 counters; rank 0 recomputes the WHOLE batch
    with the CPU oracle and requires every counter to match."""
    cmd = [sys.executable, "bench.py", "--gpus", "2"]        # workers: self-launched; process: under torchrun
    if sharding == "process":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
               "--master-addr", "127.0.0.1", "--master-port", "29545", "bench.py", "--gpus", "2"]
    d = _one_line(cmd + ["--backend", "gloo", "--objects", "30000", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0",
                         "--workload", "cfg5", "--sizes", "4096-262144", "--corrupt", "1e-4", "--verify-oracle",
                         "--sharding", sharding, "--global-table"])
    assert d["counters_equal_oracle"] is True
    assert d["config"]["launcher"] == ("self" if sharding == "workers" else "external")
    # the optional second form of the exchange: every rank ends up with the first bad unit of every object
    # of the job (minimum over ranks of tables that each hold one rank's share)
    assert d["global_table_equals_oracle"] is True
    assert d["config"]["global_table_objects"] == 2 * d["config"]["objects_per_gpu"]
    assert d["global_table_bad_objects"] == d["counters"][1]
    c = d["counters"]
    assert c[0] == 2 * d["config"]["objects_per_gpu"] and c[1] > 0 and c[3] >= c[1] and c[9] >= c[3]
    if sharding == "workers":
        assert "one key space (process ID 0)" in d["config"]["sharding"]


def test_capi_is_the_default_collective_with_one_rank():
    """Under the RCCL backend the counter sum runs through the library's own entry point
    (oval_allreduce_counters -> ncclAllReduce); the rank count in the line is the communicator's."""
    d = _one_line([sys.executable, "bench.py", "--objects", "20000", "--steps", "2", "--warmup", "1",
                   "--cpu-seconds", "0", "--no-traffic", "--no-also"])
    assert d["config"]["collective"] == "capi" and d["config"]["ranks_seen"] == 1, d["config"]


def test_capi_collective_refuses_more_ranks_than_gpus():
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29547", "bench.py", "--gpus", "2",
                        "--backend", "gloo", "--collective", "capi", "--objects", "2000", "--steps", "1",
                        "--warmup", "0", "--cpu-seconds", "0"], cwd=ROOT, capture_output=True, text=True, timeout=280)
    assert p.returncode != 0
    assert "needs one GPU per rank" in p.stderr


def test_cfg5_share_of_one_gpu_at_full_size_with_damage():
    """cfg5's per-GPU share at its real size (1.25 M objects of 4k-1M, ~583 GB,
    run as resident sub-batches through one body buffer), damaged at 1e-5 of its
    units after each fill: every unit checked, and all 12 counters equal what the
    damage plans imply (tests/test_batch.py checks that derivation against the
    oracle at sizes a CPU pass can reach)."""
    d = _one_line([sys.executable, "bench.py", "--workload", "cfg5", "--corrupt", "1e-5", "--verify-plan",
                   "--steps", "1", "--warmup", "0", "--cpu-seconds", "0"])
    assert d["counters_equal_plan"] is True
    c = d["counters"]
    assert d["config"]["bytes_per_gpu"] > 500e9 and d["config"]["sub_batches"] >= 8
    assert c[0] == d["config"]["objects_per_gpu"] and c[2] == d["config"]["units_per_gpu"]
    # ~1e-5 of 2.28 G units are damaged: half stale (all reference-visible), half by one bit flip, of
    # which 40/256 land in the five compared fields and 248/256 outside the time bytes
    assert 12000 < c[3] < 14500 and 10500 < c[6] < 12500 and 21000 < c[9] < 24000

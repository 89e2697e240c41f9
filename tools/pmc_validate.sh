#!/bin/bash
# usage: tools/pmc_validate.sh <fraction>  -- SQ counters of the validate kernel at one stale fraction (cfg2)
f=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_val_$f
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/a -- python3 $R/tools/corrupt_rates.py --fractions $f > $OUT/a.log 2>&1 || { echo "pmc failed"; tail -5 $OUT/a.log; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "validate_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    v = agg[k]
    print("%-24s launches %d  mean %.4g" % (k, len(v), sum(v) / len(v)))
PY

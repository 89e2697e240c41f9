#!/bin/bash
# usage: tools/scalar_fetch_probe.sh [table MiB]   -- FETCH_SIZE (KiB) per variant of build/scalar_fetch_probe
R=${GRAFT_REPO_ROOT:-$(pwd)}
MIB=${1:-1024}
export TMPDIR=/tmp
cd /tmp
for v in wide vector_one scalar_one scalar_pair scalar_two; do
  out=/tmp/sfp_$v
  rm -rf $out
  timeout -s KILL 90 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out -- $R/build/scalar_fetch_probe $v $MIB > $out.log 2>&1 || { echo "$v failed"; tail -3 $out.log; exit 1; }
  python3 - "$out" "$v" "$MIB" <<'PY'
import csv, glob, sys
out, v, mib = sys.argv[1], sys.argv[2], int(sys.argv[3])
tot = 0.0
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE" and v in r["Kernel_Name"]:
            tot += float(r["Counter_Value"])
print(f"{v:12s} FETCH_SIZE {tot:12.0f} KiB  = {tot / (mib * 1024):.3f} x the table's {mib * 1024} KiB")
PY
done

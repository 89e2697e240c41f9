#!/bin/bash
# usage: tools/profile_round.sh <tag> <workload> [bench args...]
# Three separate rocprofv3 runs of the same bench command (4 passes over the batch):
# kernel trace + stats, then one PMC pass each for FETCH_SIZE and WRITE_SIZE.
# Output under gpurun_out/<tag>_<workload>/; tools/traffic_summary.py condenses it.
tag=$1; wl=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${tag}_${wl}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python3 $R/bench.py --workload $wl --steps 3 --warmup 1 --cpu-seconds 0 --no-traffic --no-also $*"
timeout -s KILL 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/stats.log 2>&1 || { echo "stats run failed"; exit 1; }
timeout -s KILL 280 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $BENCH > $OUT/fetch.log 2>&1 || { echo "FETCH_SIZE run failed"; exit 1; }
timeout -s KILL 280 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $BENCH > $OUT/write.log 2>&1 || { echo "WRITE_SIZE run failed"; exit 1; }
python3 $R/tools/traffic_summary.py $OUT 4 > $OUT/summary.txt && cat $OUT/summary.txt

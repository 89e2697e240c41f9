#!/bin/bash
# usage: tools/sweep.sh <workload> "<variants>" "<bpcs>"  -- prints one line per (variant, blocks/CU)
wl=$1; shift
for v in $1; do for b in $2; do
  timeout -k 10 120 python -u bench.py --workload $wl --steps 10 --warmup 2 --variant $v --blocks-per-cu $b --cpu-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']; print('$wl variant %3s bpc %2s  total %7.1f  fill %7.1f  validate %7.1f' % (d['config']['variant'], d['config']['blocks_per_cu'], d['value'], k['fill_GBps'], k['validate_GBps']))"
done; done

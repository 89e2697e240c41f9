#!/bin/bash
# object-size sweep: tools/sweep2.sh "<variants>" "<bpcs>"
for spec in "1000000 4096-16384" "300000 32768-32768" "150000 65536-65536" "40000 262144-262144" "10000 1048576-1048576" "20000 4096-1048576" "600 16777216-16777216"; do
  set -- $spec; n=$1; sz=$2
  for v in $VARIANTS; do for b in $BPCS; do
    timeout -k 10 120 python -u bench.py --workload cfg2 --objects $n --sizes $sz --steps 8 --warmup 2 --variant $v --blocks-per-cu $b --cpu-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']; print('n %8d sizes %-18s variant %3d bpc %2d  total %7.1f  fill %7.1f  validate %7.1f  (%.1f GB)' % ($n, '$sz', d['config']['variant'], d['config']['blocks_per_cu'], d['value'], k['fill_GBps'], k['validate_GBps'], d['config']['bytes_per_gpu']/1e9))"
  done; done
done

#!/bin/bash
# Objects of one to a few units (the reference's smallest object is one unit, pattern.go:23-34), 2-4 GB batches, all
# on one box: the small-object kernels off (0), at the library's limits (default), and forced on for every shape.
#   tools/small_sweep.sh
for spec in "8000000 256-256" "8000000 512-512" "4000000 1024-1024" "4000000 256-1024" "2500000 1536-1536" "4000000 256-2048" "2000000 2048-2048" "2000000 256-4096" "1300000 3072-3072" "1000000 4096-4096" "2000000 256-8192" "1000000 4096-16384"; do
  set -- $spec
  for small in 0 default 9223372036854775807; do
    opt="--small-object-bytes $small"; [ $small = default ] && opt=""
    timeout -k 10 170 python -u bench.py --workload cfg2 --objects $1 --sizes $2 $opt --steps 10 --warmup 2 --cpu-seconds 0 --no-traffic --no-also 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']; print('n %8d sizes %-12s small-object kernels %-7s  fill %7.1f GB/s %8.4f ms  validate %7.1f GB/s %8.4f ms  (%.2f GB)  %s / %s' % ($1, '$2', {'0': 'off', 'default': 'default'}.get('$small', 'on'), k['fill_GBps'], k['fill_ms'], k['validate_GBps'], k['validate_ms'], d['config']['bytes_per_gpu']/1e9, d['roofline_fill']['kernel'], d['roofline_validate']['kernel']), flush=True)" || exit 1
  done
done

#!/bin/bash
# object-size sweep of the two passes at fixed batches of 2-9 GB, all on one box: tools/size_sweep.sh
for spec in "8000000 256-256" "8000000 512-512" "4000000 1024-1024" "4000000 256-1024" "2000000 2048-2048" "2000000 256-4096" "2000000 4096-4096" "1900000 4352-4352" "2000000 256-8192" "1000000 8192-8192" "500000 16384-16384" "125000 65536-65536" "120000 65792-65792" "8000 1048576-1048576" "512 16777216-16777216" "32 268435456-268435456" "1000000 4096-16384"; do
  set -- $spec
  timeout -k 10 170 python -u bench.py --workload cfg2 --objects $1 --sizes $2 --steps 10 --warmup 2 --cpu-seconds 0 --no-traffic --no-also 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']; print('n %8d sizes %-20s total %7.1f  fill %7.1f  validate %7.1f GB/s  (%.1f GB)  %s / %s' % ($1, '$2', d['value'], k['fill_GBps'], k['validate_GBps'], d['config']['bytes_per_gpu']/1e9, d['roofline_fill']['kernel'], d['roofline_validate']['kernel']), flush=True)" || exit 1
done

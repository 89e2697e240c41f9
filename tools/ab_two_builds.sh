# usage: bash tools/ab_two_builds.sh   -- bench.py with build/before/liboval_hip.so (OVAL_HIP_LIB) and with the tree's own build, alternating, three rounds
for i in 1 2 3; do for v in before after; do
  if [ $v = before ]; then export OVAL_HIP_LIB=$GRAFT_REPO_ROOT/build/before/liboval_hip.so; else unset OVAL_HIP_LIB; fi
  for w in "--workload cfg2" "--workload cfg2 --sizes 4096-4096" "--workload cfg5 --objects 20000"; do
    timeout -k 10 200 python bench.py --no-also --cpu-seconds 0 --no-traffic $w > /tmp/o.json 2>/dev/null || exit 1
    python -c "
import json
d=json.load(open('/tmp/o.json')); f=d['roofline_fill']; v=d['roofline_validate']
print('$v', '$w'.ljust(36), 'fill %.4f ms %.0f GB/s (%d launches)  validate %.4f ms %.0f GB/s' % (f['pass_ms'], f['achieved'], f['launches_per_step'], v['pass_ms'], v['achieved']))"
  done; done; done

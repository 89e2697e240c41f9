# usage: bash tools/ab_cfg3.sh <dir with the other build's liboval_hip.so>   -- cfg3 and cfg2 of two builds alternating on one box
other=$GRAFT_REPO_ROOT/$1/liboval_hip.so
for i in 1 2 3; do for v in other tree; do
  if [ $v = other ]; then export OVAL_HIP_LIB=$other; else unset OVAL_HIP_LIB; fi
  for w in cfg3 cfg2; do
    timeout -k 10 200 python bench.py --workload $w --no-also --cpu-seconds 0 --no-traffic > /tmp/o.json 2>/dev/null || exit 1
    python -c "
import json
d=json.load(open('/tmp/o.json')); f=d['roofline_fill']; v=d['roofline_validate']
print('$v $w: fill %.4f ms %.0f GB/s  validate %.4f ms %.0f GB/s' % (f['pass_ms'], f['achieved'], v['pass_ms'], v['achieved']), flush=True)"
  done
done; done

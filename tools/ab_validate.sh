# usage: bash tools/ab_validate.sh <dir with the other build's liboval_hip.so>
# The validate kernel of two builds, alternating on one box: clean cfg2 (bench.py) and the corruption-rate sweep.
other=$GRAFT_REPO_ROOT/$1/liboval_hip.so
for i in 1 2 3; do for v in before after; do
  if [ $v = before ]; then export OVAL_HIP_LIB=$other; lib="--lib $other"; else unset OVAL_HIP_LIB; lib=""; fi
  timeout -k 10 200 python bench.py --no-also --cpu-seconds 0 --no-traffic --no-compat > /tmp/o.json 2>/dev/null || exit 1
  python -c "
import json
d=json.load(open('/tmp/o.json')); f=d['roofline_fill']; v=d['roofline_validate']
print('$v cfg2 clean: fill %.4f ms %.0f GB/s  validate %.4f ms %.0f GB/s' % (f['pass_ms'], f['achieved'], v['pass_ms'], v['achieved']), flush=True)"
  echo "== $v"; timeout -k 10 200 python -u tools/corrupt_rates.py --fractions 0,1e-3,1e-1,1 $lib 2>/dev/null || exit 1
done; done

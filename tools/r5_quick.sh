#!/bin/bash
# GPU box: parity suite (optional -k filter as $2, "none" skips it), then the short bench line twice
set -u
tag=$1; filt=${2:-}
out=gpurun_out/$tag
mkdir -p $out
cd "$GRAFT_REPO_ROOT"
if [ "$filt" != "none" ]; then
  if [ -n "$filt" ]; then
    timeout 900 python3 -m pytest tests -m gpu -x -q -k "$filt" > $out/gpu_tests.log 2>&1
  else
    timeout 900 python3 -m pytest tests -m gpu -x -q > $out/gpu_tests.log 2>&1
  fi
  rc=$?; echo "tests rc=$rc"; tail -3 $out/gpu_tests.log
  [ $rc -ne 0 ] && exit $rc
fi
for i in 1 2; do
timeout 300 python3 bench.py --no-cpu-baseline --no-host-path --no-ragged --no-other-modes --steps 5 --profile-out $out/events$i.csv > $out/bench$i.json 2> $out/bench$i.err
python3 -c "
import json
l=json.loads(open('$out/bench$i.json').read().strip().splitlines()[-1])
sb=l.get('small_batch') or {}
print('value', l['value'], 'comp', l['compress_GBps_per_gpu'], 'decomp', l['decompress_GBps_per_gpu'], 'small', sb.get('compress_GBps'), sb.get('decompress_GBps'), 'launches', l['roofline']['launches_per_step_all_kernels'], 'err', l['max_abs_err'])
"
done
grep -E "xyz|ref_assemble|lift_axis" $out/events2.csv

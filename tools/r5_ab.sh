#!/bin/bash
# GPU box: parity suite, then A/B of environment settings on the bench volume
#   bash tools/r5_ab.sh TAG "ENV1=.. ENV2=.." "ENV=.." ...
set -u
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
cd "$GRAFT_REPO_ROOT"
timeout 900 python3 -m pytest tests -m gpu -x -q > $out/gpu_tests.log 2>&1
rc=$?
echo "tests rc=$rc"; tail -3 $out/gpu_tests.log
[ $rc -ne 0 ] && exit $rc
for envs in "$@"; do
  env $envs timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-host-path --no-ragged --no-other-modes --steps 5 2>$out/bench.err < /dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.readline())
sb=l.get('small_batch') or {}
print('$envs', 'value', l['value'], 'comp', l['compress_GBps_per_gpu'], 'decomp', l['decompress_GBps_per_gpu'], 'small', sb.get('compress_GBps'), sb.get('decompress_GBps'), 'launches', l['roofline']['launches_per_step_all_kernels'], 'top5', l['roofline']['top5_ms_per_step'])
" >> $out/ab.txt
done
cat $out/ab.txt

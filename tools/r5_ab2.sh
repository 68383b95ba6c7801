#!/bin/bash
# GPU box: A/B of environment settings on the bench volume, no test suite:  bash tools/r5_ab2.sh TAG "ENV.." ...
set -u
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
cd "$GRAFT_REPO_ROOT"
for envs in "$@"; do
  env $envs timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-host-path --no-ragged --no-other-modes --steps 5 2>$out/bench.err < /dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.readline())
sb=l.get('small_batch') or {}
print('$envs', 'value', l['value'], 'comp', l['compress_GBps_per_gpu'], 'decomp', l['decompress_GBps_per_gpu'], 'small', sb.get('compress_GBps'), sb.get('decompress_GBps'), sb.get('decoded_identical_to_big_batch'), 'err', l['max_abs_err'], 'top3', list(l['roofline']['top5_ms_per_step'].items())[:3])
" >> $out/ab.txt
done
cat $out/ab.txt

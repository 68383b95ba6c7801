#!/bin/bash
# GPU box: the short bench line under environment settings, alternating:  bash tools/r5_env.sh TAG "A=1" "" "A=1" ""
set -u
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
cd "$GRAFT_REPO_ROOT"
i=0
for envs in "$@"; do
  i=$((i+1))
  env $envs timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-host-path --no-ragged --no-other-modes --steps 5 --profile-out $out/events$i.csv 2>$out/bench$i.err < /dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.readline())
sb=l.get('small_batch') or {}
print('[$envs]', 'value', l['value'], 'comp', l['compress_GBps_per_gpu'], 'decomp', l['decompress_GBps_per_gpu'], 'small', sb.get('compress_GBps'), sb.get('decompress_GBps'), 'err', l['max_abs_err'])
"
  grep -E "xyz_inv|ref_assemble|lift_axis<false" $out/events$i.csv
done

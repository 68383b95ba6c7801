#!/bin/bash
# A/B of environment settings on the bench volume (GPU box): bash tools/ab.sh OUT "A=0" "SPERR_HIP_X=1 SPERR_HIP_Y=2" ...
# (each argument is one run's environment; results are appended to OUT as they come)
out=$1; shift
for envs in "$@"; do
  env $envs timeout -k 10 150 python bench.py --no-cpu-baseline --no-host-path --no-ragged --no-other-modes --steps 5 2>/dev/null < /dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline())
sb=l.get('small_batch') or {}
print('$envs', 'value', l['value'], 'comp', l['compress_GBps_per_gpu'], 'decomp', l['decompress_GBps_per_gpu'], 'small', sb.get('compress_GBps'), sb.get('decompress_GBps'))
" >> $out
done

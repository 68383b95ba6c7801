#!/bin/bash
# A/B of environment settings on the host-resident farm path (GPU box): bash tools/ab_host.sh OUT "ENV..." ...
out=$1; shift
for envs in "$@"; do
  env $envs timeout -k 10 200 python bench.py --no-cpu-baseline --no-ragged --no-other-modes --steps 3 2>/dev/null < /dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline())
hp=l.get('host_path') or {}; hw=l.get('host_path_pwe') or {}; sb=l.get('small_batch') or {}
print('$envs', 'value', l['value'], 'host comp', hp.get('compress_GBps'), 'decomp', hp.get('decompress_GBps'), 'pwe', hw.get('compress_GBps'), hw.get('decompress_GBps'), 'small', sb.get('compress_GBps'), sb.get('decompress_GBps'))
" >> $out
done

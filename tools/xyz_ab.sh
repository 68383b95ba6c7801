#!/bin/bash
# fused x-y-z lifting pass on / off on the bench volume (GPU box)
for v in 1 0; do
  SPERR_HIP_LIFT_XYZ=$v timeout -k 10 200 python bench.py --no-cpu-baseline --no-host-path --no-ragged --no-other-modes --steps 5 --profile-out gpurun_out/xyz_$v.csv 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline())
sb=l.get('small_batch') or {}
print('XYZ=$v', 'value', l['value'], 'comp', l['compress_GBps_per_gpu'], 'decomp', l['decompress_GBps_per_gpu'], 'small', sb.get('compress_GBps'), sb.get('decompress_GBps'), 'err', l['max_abs_err'], l['container_len_exact'])
"
  grep -i "lift" gpurun_out/xyz_$v.csv
done

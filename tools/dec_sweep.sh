#!/bin/bash
# Decoder knobs swept on the bench volume: bash tools/dec_sweep.sh  (GPU box)
run() {
  env "$@" timeout -k 10 150 python bench.py --no-cpu-baseline --no-host-path --no-ragged --no-other-modes --steps 5 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline())
sb=l.get('small_batch') or {}
print('$*', 'value', l['value'], 'comp', l['compress_GBps_per_gpu'], 'decomp', l['decompress_GBps_per_gpu'], 'small', sb.get('compress_GBps'), sb.get('decompress_GBps'), 'top', l['roofline']['kernel'], l['roofline']['kernel_ms_per_step'], l['roofline']['kernel_busy_ms_per_step'])
"
}
run A=0
run SPERR_HIP_HI_HOP2=0
run SPERR_HIP_HI_EXTRA=0
run SPERR_HIP_HI_HOP2=0 SPERR_HIP_HI_EXTRA=0
run SPERR_HIP_HI_HOP2=0 SPERR_HIP_HI_WGS=192
run SPERR_HIP_LIVE_CHECK=0

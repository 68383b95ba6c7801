#!/bin/bash
# Decoder knobs swept on the bench volume: bash tools/dec_sweep.sh  (GPU box)
run() {
  env "$@" timeout -k 10 150 python bench.py --no-cpu-baseline --no-host-path --no-ragged --no-other-modes --steps 5 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline())
sb=l.get('small_batch') or {}
print('$*', 'value', l['value'], 'comp', l['compress_GBps_per_gpu'], 'decomp', l['decompress_GBps_per_gpu'], 'small', sb.get('compress_GBps'), sb.get('decompress_GBps'))
"
}
run A=0
run SPERR_HIP_HI_WGS=128
run SPERR_HIP_HI_WGS=224
run SPERR_HIP_HI_WGS=320
run SPERR_HIP_L1_WGS=1024 SPERR_HIP_L0_WGS=768
run SPERR_HIP_L1_WGS=512 SPERR_HIP_L0_WGS=384
run SPERR_HIP_HI_AHEAD=256
run SPERR_HIP_HI_AHEAD=512

#!/bin/bash
for k in 9 5 4 3; do
  echo "KCAP=$k $(SPERR_HIP_HI_KCAP=$k timeout -k 5 100 python tools/hi_stamps.py 512 2>/dev/null | head -6 | tr '\n' ' ')"
  SPERR_HIP_HI_KCAP=$k timeout -k 10 150 python bench.py --no-cpu-baseline --no-host-path --no-ragged --no-other-modes --steps 5 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline())
print('   value', l['value'], 'decomp', l['decompress_GBps_per_gpu'], 'small', l['small_batch']['decompress_GBps'], l['small_batch']['decompress_ms'], 'hi', l['roofline']['kernel'], l['roofline']['kernel_ms_per_step'], l['max_abs_err'])
"
done

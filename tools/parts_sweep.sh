#!/bin/bash
# compress parts / decode sub-batches swept on the bench volume (GPU box)
run() {
  env "$@" timeout -k 10 150 python bench.py --no-cpu-baseline --no-host-path --no-ragged --no-other-modes --no-small-batch --steps 5 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline())
print('$*', 'value', l['value'], 'comp', l['compress_GBps_per_gpu'], 'decomp', l['decompress_GBps_per_gpu'])
"
}
run SPERR_HIP_ENC_PARTS=1
run SPERR_HIP_ENC_PARTS=2
run SPERR_HIP_ENC_PARTS=3
run SPERR_HIP_ENC_PARTS=4
run SPERR_HIP_SUBSTREAMS=1
run SPERR_HIP_SUBSTREAMS=3
run SPERR_HIP_SUBSTREAMS=4

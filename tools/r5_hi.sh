#!/bin/bash
# GPU box: parity of the decoders, then k_lis_hi's stamps and the small batch
set -u
tag=$1
out=gpurun_out/$tag
mkdir -p $out
cd "$GRAFT_REPO_ROOT"
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_golden_big.py tests/test_gpu_fuzz.py -m gpu -x -q > $out/gpu_tests.log 2>&1
rc=$?
echo "tests rc=$rc"; tail -3 $out/gpu_tests.log
[ $rc -ne 0 ] && exit $rc
timeout 200 python3 tools/hi_stamps.py 512 > $out/hi_stamps.txt 2>&1
cat $out/hi_stamps.txt
timeout 200 python3 tools/small_batch.py 512 x > $out/small_batch.txt 2>&1
cat $out/small_batch.txt
timeout 200 python3 bench.py --no-cpu-baseline --no-host-path --no-ragged --no-other-modes --steps 5 2>$out/bench.err | python3 -c "
import json,sys
l=json.loads(sys.stdin.readline())
sb=l.get('small_batch') or {}
print('value', l['value'], 'comp', l['compress_GBps_per_gpu'], 'decomp', l['decompress_GBps_per_gpu'], 'small', sb.get('compress_GBps'), sb.get('decompress_GBps'), 'top5', l['roofline']['top5_ms_per_step'])
" | tee $out/bench.txt

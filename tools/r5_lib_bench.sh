#!/bin/bash
# GPU box: the short bench line and k_lis_hi's stamps under variant libraries:  bash tools/r5_lib_bench.sh lib1.so lib2.so ...
set -u
cd "$GRAFT_REPO_ROOT"
for lib in "$@"; do
  echo "== $lib"
  SPERR_HIP_LIB=$lib timeout 200 python3 bench.py --no-cpu-baseline --no-host-path --no-ragged --no-other-modes --steps 5 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.readline()); sb=l.get('small_batch') or {}
print('value', l['value'], 'comp', l['compress_GBps_per_gpu'], 'decomp', l['decompress_GBps_per_gpu'], 'small', sb.get('compress_GBps'), sb.get('decompress_GBps'))"
  SPERR_HIP_LIB=$lib timeout 100 python3 tools/hi_stamps.py 1024 2>&1 | grep -E "table phase"
done

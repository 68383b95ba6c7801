#!/bin/bash
# GPU box: per-kernel sums of the 8-chunk batch under variant libraries:  bash tools/r5_lib_ab.sh TAG lib1.so lib2.so ...
set -u
tag=$1; shift
mkdir -p gpurun_out/$tag
cd "$GRAFT_REPO_ROOT"
for lib in "$@"; do
  echo "== $lib"
  SPERR_HIP_LIB=$lib timeout 200 python3 tools/small_batch.py 512 1 2>&1 | grep -E "compress |split_emit|emit_pixels|list_apply|decompress "
done

#!/bin/bash
# GPU box: small-batch time and k_lis_hi stamps for library variants: bash tools/r5_var.sh TAG lib1 lib2 ...
set -u
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
cd "$GRAFT_REPO_ROOT"
for lib in "$@"; do
  for rep in 1 2; do
    SPERR_HIP_LIB=$lib timeout 200 python3 tools/small_batch.py 512 2>/dev/null | grep -E "^(de)?compress" | sed "s|^|$lib |" >> $out/var.txt
  done
  SPERR_HIP_LIB=$lib timeout 200 python3 tools/hi_stamps.py 512 2>/dev/null | grep -E "chain   |on-chain" | sed "s|^|$lib |" >> $out/var.txt
done
cat $out/var.txt

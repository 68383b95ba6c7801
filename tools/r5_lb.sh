#!/bin/bash
# GPU box: decoder parity, then A/B of the look-back with the diagnostics build
set -u
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
cd "$GRAFT_REPO_ROOT"
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_golden_big.py tests/test_gpu_fuzz.py -m gpu -x -q > $out/gpu_tests.log 2>&1
rc=$?
echo "tests rc=$rc"; tail -3 $out/gpu_tests.log
[ $rc -ne 0 ] && exit $rc
bash tools/r5_ab2.sh $tag "$@"

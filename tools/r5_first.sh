#!/bin/bash
# round 5, first GPU call: the suite, the bench line, kernel traces of the 64-chunk step and of an 8-chunk batch
set -u
out=gpurun_out/r5a
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 900 python3 -m pytest tests -m gpu -x -q > $out/gpu_tests.log 2>&1
echo "tests rc=$?"; tail -2 $out/gpu_tests.log
timeout 600 python3 bench.py > $out/bench.json 2> $out/bench.err
echo "bench rc=$?"
args="bench.py --size 1024 --steps 1 --warmup 1 --no-cpu-baseline --no-host-path --no-ragged --no-other-modes --no-small-batch"
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace64 -o run -- python3 $args > $out/trace64.log 2>&1
echo "trace64 rc=$?"
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace8 -o run -- python3 tools/small_batch.py 512 > $out/trace8.log 2>&1
echo "trace8 rc=$?"
cut -c1-600 $out/bench.json

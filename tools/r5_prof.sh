#!/bin/bash
# GPU box: per-kernel event table of the bench step + kernel trace (plane table)
set -u
tag=$1
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 300 python3 bench.py --no-cpu-baseline --no-host-path --no-ragged --no-other-modes --steps 5 --profile-out $out/events.csv > $out/bench.json 2> $out/bench.err
echo "bench rc=$?"
args="bench.py --size 1024 --steps 1 --warmup 1 --no-cpu-baseline --no-host-path --no-ragged --no-other-modes --no-small-batch"
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace64 -o run -- python3 $args > $out/trace64.log 2>&1
echo "trace64 rc=$?"
python3 tools/plane_table2.py $out/trace64/run_kernel_trace.csv > $out/plane_table64.txt 2>&1
head -30 $out/events.csv

#!/bin/bash
# Runs on the GPU box (gpurun): kernel-trace statistics and HBM counters of the bench workload.
#   bash tools/collect_profiles.sh <tag>         -> gpurun_out/prof_<tag>/{trace,fetch,write}/...
# The counters are collected in their own passes (never together with a trace domain); every
# profiler run is bounded by `timeout` (a wedged profiler must not eat the GPU budget).
set -u
tag=${1:-r1}
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
python3 -c "from sperr_amd.srchash import bench_path_hash as h; print(h())" > $out/source_sha16.txt
args="bench.py --size 1024 --steps 1 --warmup 1 --no-cpu-baseline --no-host-path --no-ragged --no-other-modes --no-small-batch"
timeout 600 python3 bench.py --size 1024 --steps 5 --warmup 1 --no-ragged --no-other-modes --profile-out $out/engine_events.csv > $out/bench.json 2> $out/bench.err
echo "bench rc=$?"
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o run -- python3 $args > $out/trace.log 2>&1
echo "trace rc=$?"
timeout 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o run -- python3 $args > $out/fetch.log 2>&1
echo "fetch rc=$?"
# the WRITE_SIZE pass runs on its own box: tools/collect_write_pass.sh (a third profiler session
# on the same box has wedged before the workload's first kernel)
tail -1 $out/bench.json | cut -c1-400

#!/bin/bash
# The FETCH_SIZE counter pass on its own box (profiler sessions after the first of one box have
# wedged more than once): bash tools/collect_fetch_pass.sh <tag>
set -u
tag=${1:-r1}
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o run -- python3 bench.py --size 1024 --steps 1 --warmup 1 --no-cpu-baseline --no-host-path > $out/fetch.log 2>&1
echo "fetch rc=$?"

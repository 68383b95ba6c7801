#!/bin/bash
# The WRITE_SIZE counter pass on its own box (as the third profiler run of one session it has
# wedged twice before the first kernel of the workload): bash tools/collect_write_pass.sh <tag>
set -u
tag=${1:-r1}
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
python3 -c "from sperr_amd.srchash import bench_path_hash as h; print(h())" > $out/source_sha16.txt
timeout 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o run -- python3 bench.py --size 1024 --steps 1 --warmup 1 --no-cpu-baseline --no-host-path --no-ragged --no-other-modes --no-small-batch > $out/write.log 2>&1
echo "write rc=$?"

#!/bin/bash
set -u
out=gpurun_out/r5_pmcprobe
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
python3 -c "import torch; print(torch.cuda.is_available())"
args="bench.py --size 1024 --steps 1 --warmup 1 --no-cpu-baseline --no-host-path --no-ragged --no-other-modes --no-small-batch"
t0=$(date +%s)
timeout -k 10 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetchB -o run -- python3 $args > $out/fetchB.log 2>&1
echo "product rc=$? $(( $(date +%s) - t0 )) s"

#!/bin/bash
# GPU box: the state at HEAD -- suite, event table + kernel trace (plane tables at 64 and 8 chunks), list-kernel stamps
set -u
tag=$1
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
if [ "${2:-tests}" = tests ]; then
  timeout 900 python3 -m pytest tests -m gpu -x -q > $out/gpu_tests.log 2>&1
  rc=$?; echo "tests rc=$rc"; tail -2 $out/gpu_tests.log
  [ $rc -ne 0 ] && exit $rc
fi
timeout 300 python3 bench.py --no-cpu-baseline --no-host-path --no-ragged --no-other-modes --steps 5 --profile-out $out/events.csv > $out/bench.json 2> $out/bench.err
echo "bench rc=$?"
python3 -c "
import json
l=json.loads(open('$out/bench.json').read().strip().splitlines()[-1])
sb=l.get('small_batch') or {}
print('value', l['value'], 'comp', l['compress_GBps_per_gpu'], 'decomp', l['decompress_GBps_per_gpu'], 'small', sb.get('compress_GBps'), sb.get('decompress_GBps'), 'launches', l['roofline']['launches_per_step_all_kernels'])
"
args="bench.py --size 1024 --steps 1 --warmup 1 --no-cpu-baseline --no-host-path --no-ragged --no-other-modes --no-small-batch"
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace64 -o run -- python3 $args > $out/trace64.log 2>&1
echo "trace64 rc=$?"
python3 tools/plane_table2.py $out/trace64/run_kernel_trace.csv > $out/plane_table64.txt 2>&1
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace8 -o run -- python3 tools/small_batch.py 512 > $out/trace8.log 2>&1
echo "trace8 rc=$?"
python3 tools/plane_table2.py $out/trace8/run_kernel_trace.csv > $out/plane_table8.txt 2>&1
timeout 120 python3 tools/hi_stamps.py 1024 > $out/hi_stamps64.txt 2>&1
timeout 120 python3 tools/hi_stamps.py 512 > $out/hi_stamps8.txt 2>&1
timeout 120 python3 tools/small_batch.py 512 1 > $out/small8.txt 2>&1
rm -f $out/trace64/*agent_info* $out/trace8/*agent_info*
head -40 $out/events.csv

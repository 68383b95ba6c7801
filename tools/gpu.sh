#!/bin/bash
# Everything a GPU box is asked to do during development, one script (round 6: replaces tools/ab.sh and the
# seventeen tools/r5_*.sh, which differed by a line or two).  Runs from the repo root on the box:
#
#   bash tools/gpu.sh TAG STEP [STEP ...]          -> gpurun_out/TAG/...
#
# steps, run in the order given; a failing step ends the call (no GPU step is started after one that failed or
# timed out):
#   tests[:EXPR]        the -m gpu suite (pytest -k EXPR when given)        -> gpu_tests.log
#   short[:ENV..]       the short bench line (no CPU baseline, host path, ragged volume, other modes) under the
#                       environment ENV ("A=1 B=2"; may be empty), with the engine's event table -> short.txt, eventsN.csv
#   lib:PATH            the same under SPERR_HIP_LIB=PATH (a variant library of tools/build_variant.sh)
#   full                the default bench line, as the driver runs it       -> bench.json (+ a digest on stdout)
#   trace[:N]           rocprofv3 kernel trace + stats of one bench step (N = 64 chunks: the 1024^3 volume; N = 8:
#                       tools/small_batch.py 512) and the decoder's plane table -> traceN/, plane_tableN.txt
#   ragged              kernel trace of the 1000^3 volume's decompression per queue -> ragged.txt
#   stamps[:SIZE]       k_lis_hi's per-region cycle counters (tools/hi_stamps.py SIZE) -> hi_stampsSIZE.txt
#   py:SCRIPT[:ARGS]    python3 tools/SCRIPT ARGS                                    -> SCRIPT.txt
set -u
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname $0)/..}"
short_args="--no-cpu-baseline --no-host-path --no-ragged --no-other-modes --steps 5"
digest='
import json,sys
l=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
sb=l.get("small_batch") or {}
r=l["roofline"]
print(sys.argv[2], "value", l["value"], "comp", l["compress_GBps_per_gpu"], "decomp", l["decompress_GBps_per_gpu"],
      "small", sb.get("compress_GBps"), sb.get("decompress_GBps"), sb.get("decoded_identical_to_big_batch"),
      "launches", r["launches_per_step_all_kernels"], "err", l["max_abs_err"], "top3", list(r["top5_ms_per_step"].items())[:3])
'
n=0
for step in "$@"; do
  kind=${step%%:*}; arg=""; [ "$step" != "$kind" ] && arg=${step#*:}
  n=$((n+1))
  case $kind in
    tests)
      if [ -n "$arg" ]; then timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q -k "$arg" > $out/gpu_tests.log 2>&1
      else timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $out/gpu_tests.log 2>&1; fi
      rc=$?; echo "tests rc=$rc"; tail -3 $out/gpu_tests.log ;;
    short|lib)
      envs="$arg"; [ $kind = lib ] && envs="SPERR_HIP_LIB=$arg"
      env $envs timeout -k 10 300 python3 bench.py $short_args --profile-out $out/events$n.csv > $out/short$n.json 2> $out/short$n.err < /dev/null
      rc=$?; [ $rc -eq 0 ] && python3 -c "$digest" $out/short$n.json "[$envs]" | tee -a $out/short.txt ;;
    full)
      timeout -k 10 1000 python3 bench.py > $out/bench.json 2> $out/bench.err
      rc=$?; echo "bench rc=$rc"; [ $rc -eq 0 ] && tail -1 $out/bench.json | cut -c1-1500 ;;
    trace)
      N=${arg:-64}
      if [ "$N" = 8 ]; then prog="tools/small_batch.py 512"
      else prog="bench.py --size 1024 --steps 1 --warmup 1 --no-cpu-baseline --no-host-path --no-ragged --no-other-modes --no-small-batch"; fi
      timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace$N -o run -- python3 $prog > $out/trace$N.log 2>&1
      rc=$?; echo "trace$N rc=$rc"
      python3 tools/plane_table2.py $out/trace$N/run_kernel_trace.csv > $out/plane_table$N.txt 2>&1
      rm -f $out/trace$N/*agent_info* ;;
    ragged)
      timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/ragged_trace -o run -- python3 tools/ragged_1000.py > $out/ragged.log 2>&1
      rc=$?; tail -2 $out/ragged.log
      python3 tools/trace_queues.py $out/ragged_trace/run_kernel_trace.csv > $out/ragged.txt 2>&1; head -12 $out/ragged.txt ;;
    stamps)
      S=${arg:-1024}
      timeout -k 10 200 python3 tools/hi_stamps.py $S > $out/hi_stamps$S.txt 2>&1
      rc=$?; tail -25 $out/hi_stamps$S.txt ;;
    py)
      script=${arg%%:*}; pargs=""; [ "$arg" != "$script" ] && pargs=${arg#*:}
      timeout -k 10 600 python3 tools/$script $pargs > $out/${script%.py}$n.txt 2>&1
      rc=$?; tail -30 $out/${script%.py}$n.txt ;;
    *) echo "unknown step $step"; rc=64 ;;
  esac
  if [ $rc -ne 0 ]; then echo "step '$step' failed (rc=$rc): stopping"; exit $rc; fi
done

# decompression rate of the bench volume over the workgroup counts of the GPU-wide list kernels
for l0 in ${L0:-384 512 640}; do
  for l1 in ${L1:-512 768 1024}; do
    SPERR_HIP_L0_WGS=$l0 SPERR_HIP_L1_WGS=$l1 timeout -k 10 120 python bench.py --no-cpu-baseline --no-host-path --no-ragged --no-other-modes --steps 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('l0 $l0 l1 $l1', d['value'], d['decompress_GBps_per_gpu'])"
  done
done

for wgs in ${WGS:-160 192 224 256}; do
  for ah in ${AH:-256 512 768}; do
    SPERR_HIP_HI_WGS=$wgs SPERR_HIP_HI_AHEAD=$ah timeout -k 10 120 python bench.py --no-cpu-baseline --no-host-path --no-ragged --no-other-modes --steps 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wgs $wgs ahead $ah', d['value'], d['decompress_GBps_per_gpu'])"
  done
done

import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print({k:d[k] for k in ('ms_per_step','value','compress_GBps_per_gpu','decompress_GBps_per_gpu')})

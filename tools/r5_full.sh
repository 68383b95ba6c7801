#!/bin/bash
# GPU box: the default bench line (everything), kept as a record
set -u
tag=$1
out=gpurun_out/$tag
mkdir -p $out
cd "$GRAFT_REPO_ROOT"
timeout 900 python3 bench.py > $out/bench.json 2> $out/bench.err
echo "bench rc=$?"
python3 - <<PY
import json
d=json.loads(open("$out/bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","compress_GBps_per_gpu","decompress_GBps_per_gpu")})
print("roofline", {k:d["roofline"][k] for k in ("kernel","frac","step_frac","compress_frac","decompress_frac","launches_per_step_all_kernels","traffic")})
print("cpu", {k:d["cpu_baseline"][k] for k in ("value","cores","cpu_quota","compress_GBps","decompress_GBps","cfs_throttle")})
for k in ("host_path","host_path_pwe"):
    h=d[k]; print(k, {x:h.get(x) for x in ("compress_GBps","decompress_GBps","pageable_compress_GBps","pageable_decompress_GBps","cfs_throttle")})
print("small", d["small_batch"]); print("ceiling", d["strong_ceiling_from_small_batch"]); print("farm", d["farm_threads"])
print("ragged", d["ragged_volume"]); print("other", d["other_modes"])
PY

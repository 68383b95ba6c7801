"""Times the fused x-y-z lifting kernels of one compress + decompress of the bench volume (engine events)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch

from sperr_amd.api import SperrHip
from sperr_amd.synth import turbulence_torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
eng = SperrHip()
vol = turbulence_torch((n, n, n), "cuda")
out = torch.empty_like(vol)
cbuf = torch.empty(eng.max_compressed_size(vol.shape, (256,) * 3, 2.0), dtype=torch.uint8, device="cuda")
for _ in range(2):
    s = eng.compress(vol, (256, 256, 256), 2.0, out=cbuf)
    eng.decompress(s, True, out=out, shape_zyx=vol.shape)
torch.cuda.synchronize()
import time
tc, td = [], []
for _ in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s = eng.compress(vol, (256, 256, 256), 2.0, out=cbuf)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    eng.decompress(s, True, out=out, shape_zyx=vol.shape)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    tc.append(t1 - t0); td.append(t2 - t1)
print(f"lib={os.path.basename(os.environ.get('SPERR_HIP_LIB', 'default'))} wall: compress {min(tc) * 1e3:.2f} ms {vol.numel() * 4 / min(tc) / 1e9:.1f} GB/s   "
      f"decompress {min(td) * 1e3:.2f} ms {vol.numel() * 4 / min(td) / 1e9:.1f} GB/s")
eng.profile(True)
for _ in range(3):
    s = eng.compress(vol, (256, 256, 256), 2.0, out=cbuf)
    eng.decompress(s, True, out=out, shape_zyx=vol.shape)
torch.cuda.synchronize()
eng.profile(False)
rep = eng.profile_report(with_sum=True)
tag = os.environ.get("SPERR_HIP_XYZ_DBG", "0")
for k, v in sorted(rep.items(), key=lambda kv: -kv[1][2]):
    if "lift" in k or len(sys.argv) > 2:
        print(f"dbg={tag} {k:28s} sum {v[2] / 3:8.3f} ms/step  {v[1] // 3:4d} launches  avg {v[2] / max(1, v[1]):7.3f} ms")

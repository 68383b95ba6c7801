"""bench.py's ragged_volume alone: 1000^3 fp32 in 256^3 chunks at 2 bpp, compress / decompress, per-kernel sums."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from sperr_amd.api import SperrHip
from sperr_amd.synth import turbulence_torch
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
eng = SperrHip()
rv = turbulence_torch((R, R, R), "cuda", seed=7)
rs = eng.compress(rv, (256, 256, 256), 2.0).clone()
ro = eng.decompress(rs, True)
torch.cuda.synchronize()
bc, bd = 1e9, 1e9
for _ in range(3):
    torch.cuda.synchronize(); a = time.perf_counter()
    eng.compress(rv, (256, 256, 256), 2.0)
    torch.cuda.synchronize(); b = time.perf_counter()
    ro = eng.decompress(rs, True)
    torch.cuda.synchronize(); c = time.perf_counter()
    bc, bd = min(bc, b - a), min(bd, c - b)
print("%d^3: compress %.1f ms %.1f GB/s   decompress %.1f ms %.1f GB/s   max err %.4g" %
      (R, bc * 1e3, rv.numel() * 4 / bc / 1e9, bd * 1e3, rv.numel() * 4 / bd / 1e9,
       float((ro.double() - rv.double()).abs().max())), flush=True)
if os.environ.get("PROF"):
    eng.profile(True); eng.decompress(rs, True); torch.cuda.synchronize(); eng.profile(False)
    rep = eng.profile_report(with_sum=True)
    for k, v in sorted(rep.items(), key=lambda kv: -kv[1][2])[:8]:
        print("   %-28s %8.3f ms %5d launches" % (k, v[2], v[1]))

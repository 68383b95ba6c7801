"""Compression of a 1000^3 volume in 256^3 chunks (eight shape groups side by side): wall time and per-kernel sums."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.getcwd())
import torch
from sperr_amd.api import SperrHip
from sperr_amd.synth import turbulence_torch
eng = SperrHip()
rv = turbulence_torch((1000, 1000, 1000), "cuda", seed=7)
for _ in range(2):
    eng.compress(rv, (256, 256, 256), 2.0)
torch.cuda.synchronize()
ts = []
for _ in range(3):
    torch.cuda.synchronize(); a = time.perf_counter()
    eng.compress(rv, (256, 256, 256), 2.0)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - a)
print("compress %.1f ms" % (min(ts) * 1e3))
eng.profile(True); eng.compress(rv, (256, 256, 256), 2.0); torch.cuda.synchronize(); eng.profile(False)
rep = eng.profile_report(with_sum=True)
tot = sum(v[2] for v in rep.values())
print("kernel sum %.1f ms in %d launches" % (tot, sum(v[1] for v in rep.values())))
for k, v in sorted(rep.items(), key=lambda kv: -kv[1][2])[:12]:
    print("   %-28s %8.3f ms %5d launches" % (k, v[2], v[1]))

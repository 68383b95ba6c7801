import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from sperr_amd.api import SperrHip
from sperr_amd.synth import turbulence_torch
eng = SperrHip()
dv = turbulence_torch((1300, 600, 700), "cuda")
got = eng.compress(dv, (128, 128, 128), 2.0).clone()
eng.decompress(got, True); torch.cuda.synchronize()
ts=[]
for _ in range(3):
    t=time.time(); back = eng.decompress(got, True); torch.cuda.synchronize(); ts.append(time.time()-t)
print(os.environ.get("TAG"), "decompress %.3f s" % min(ts), flush=True)
if os.environ.get("PROF"):
    eng.profile(True); eng.decompress(got, True); torch.cuda.synchronize(); eng.profile(False)
    rep = eng.profile_report(with_sum=True)
    for k, v in sorted(rep.items(), key=lambda kv: -kv[1][2])[:8]:
        print("   %-28s %8.3f ms %5d launches" % (k, v[2], v[1]))

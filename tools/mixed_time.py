"""Decode time of chunks whose lists mix set shapes (k_lis_mx; SPERR_HIP_LIS_MIXED=0: the serial
walk k_lis_walk):  python tools/mixed_time.py [edge ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from sperr_amd.api import SperrHip
from sperr_amd.synth import turbulence_torch

edges = [int(a) for a in sys.argv[1:]] or [250]
eng = SperrHip()
for n in edges:
    v = turbulence_torch((n, n, n), "cuda")
    c = eng.compress(v, (n, n, n), 2.0)
    for i in range(2):
        torch.cuda.synchronize()
        t = time.time()
        eng.profile(True)
        out = eng.decompress(c, True)
        torch.cuda.synchronize()
        rep = eng.profile_report()
        top = sorted(rep.items(), key=lambda kv: -kv[1][0])[:4]
        print("%d^3 chunk, 2 bpp, decode %d: %.3f s" % (n, i, time.time() - t),
              [(k, round(x[0], 1)) for k, x in top], flush=True)
    err = float((out.double() - v.double()).abs().max())
    print("   max abs err %.4g" % err, flush=True)

"""Compress time of a volume the chunk size does not divide (shape groups side by side or one after the
other: SPERR_HIP_ENC_GROUPS=0): python tools/ragged_compress_time.py [edge] [chunk]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from sperr_amd.api import SperrHip
from sperr_amd.synth import turbulence_torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
ch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
eng = SperrHip()
v = turbulence_torch((n, n, n), "cuda", seed=7)
for i in range(3):
    torch.cuda.synchronize()
    t = time.time()
    c = eng.compress(v, (ch, ch, ch), 2.0)
    t1 = time.time()
    torch.cuda.synchronize()
    t2 = time.time()
    print("compress %d: %.1f ms (host returned after %.1f ms), %.1f GB/s" % (i, (t2 - t) * 1e3, (t1 - t) * 1e3, v.numel() * 4 / (t2 - t) / 1e9), flush=True)

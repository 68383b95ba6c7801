"""Per-kernel times of decoding a volume that the chunk size does not divide (several shape groups)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from sperr_amd.api import SperrHip
from sperr_amd.synth import turbulence_torch

eng = SperrHip()
v = turbulence_torch((1300, 600, 700), "cuda")
c = eng.compress(v, (128, 128, 128), 2.0)
eng.decompress(c, True)
torch.cuda.synchronize()
eng.profile(True)
t = time.time()
eng.decompress(c, True)
torch.cuda.synchronize()
print("decode %.3f s" % (time.time() - t))
rep = eng.profile_report(with_sum=True)
for k, x in sorted(rep.items(), key=lambda kv: -kv[1][0])[:8]:
    print("  %-28s busy %8.1f ms  sum %8.1f ms  launches %d" % (k, x[0], x[2], x[1]))

"""Per-kernel time of one 999 x 999 slice through sperr_comp_2d / sperr_decomp_2d (engine events):
python tools/slice_profile.py [mode quality]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from sperr_amd.api import SperrHip
from sperr_amd.synth import turbulence

mode = int(sys.argv[1]) if len(sys.argv) > 1 else 2
q = float(sys.argv[2]) if len(sys.argv) > 2 else 90.0
eng = SperrHip()
img = torch.from_numpy(turbulence((1, 999, 999))[0]).cuda()
s = eng.compress_2d(img, q, mode=mode).clone()
eng.decompress_2d(s, (999, 999), True)
for what in ("compress", "decompress"):
    torch.cuda.synchronize()
    eng.profile(True)
    if what == "compress":
        eng.compress_2d(img, q, mode=mode)
    else:
        eng.decompress_2d(s, (999, 999), True)
    torch.cuda.synchronize()
    rep = eng.profile_report()
    tot = sum(x[0] for x in rep.values())
    print("%s: %.2f ms in %d launches" % (what, tot, sum(x[1] for x in rep.values())))
    for k, x in sorted(rep.items(), key=lambda kv: -kv[1][0])[:12]:
        print("   %-28s %8.3f ms %5d launches" % (k, x[0], x[1]))

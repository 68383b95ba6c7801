"""One block of the 1000^3 synthetic field as a single chunk against the reference (debugging aid):
python tests/tools/check_block.py z0 y0 x0 dz dy dx"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch

from oracle import pyoracle
from sperr_amd.api import SperrHip
from sperr_amd.synth import turbulence_torch

z0, y0, x0, dz, dy, dx = [int(a) for a in sys.argv[1:7]]
eng = SperrHip()
impl = pyoracle.Ref() if pyoracle.have_ref() else pyoracle.Oracle()
v = turbulence_torch((1000, 1000, 1000), "cuda", seed=7)[z0:z0 + dz, y0:y0 + dy, x0:x0 + dx].contiguous()
hv = v.cpu().numpy()
c = eng.compress(v, (dx, dy, dz), 2.0).clone()
want = impl.comp_3d(hv, (dx, dy, dz), 1, 2.0, nthreads=1)
print("container identical", bytes(c.cpu().numpy()) == want, len(want))
d = eng.decompress(c, True).cpu().numpy()
r = impl.decomp_3d(want, True, nthreads=1)
bad = np.argwhere(d.view(np.uint32) != r.view(np.uint32))
print("decode identical", bad.shape[0] == 0, "differing", bad.shape[0],
      "max err gpu %.4g ref %.4g" % (float(np.abs(d.astype(np.float64) - hv).max()), float(np.abs(r.astype(np.float64) - hv).max())))
if bad.shape[0] and os.environ.get("BISECT"):
    orc = pyoracle.Oracle()
    lo, hi = 0, 100
    while hi - lo > 1:
        mid = (lo + hi) // 2
        part = orc.trunc_3d(want, mid)
        dd = eng.decompress(torch.from_numpy(np.frombuffer(part, dtype=np.uint8).copy()).cuda(), True).cpu().numpy()
        rr = impl.decomp_3d(part, True, nthreads=1)
        ok = np.array_equal(dd.view(np.uint32), rr.view(np.uint32))
        print("cut %d %% (%d bytes): %s" % (mid, len(part), "same" if ok else "DIFFERENT"), flush=True)
        if ok:
            lo = mid
        else:
            hi = mid

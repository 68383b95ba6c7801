"""How many 64-sample words of a 256^3 chunk's decoded coefficients hold no significant sample (bench volume, 2 bpp)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from sperr_amd.api import SperrHip
from sperr_amd.synth import turbulence_torch
eng = SperrHip()
vol = turbulence_torch((256, 256, 256), "cuda")
s = eng.compress(vol, (256, 256, 256), 2.0).cpu().numpy()
# one chunk: 14 + 4 bytes of container header, 17 of conditioner, then the SPECK stream
coef, sign = eng.speck3d_decode(bytes(s[18 + 17:]), (256, 256, 256))[:2]
c = np.asarray(coef).reshape(-1)
nz = c != 0
w = nz.reshape(-1, 64).any(axis=1)
print("significant samples %.2f %%, words with one %.2f %%, rows of 256 with one %.2f %%" % (100 * nz.mean(), 100 * w.mean(), 100 * nz.reshape(-1, 256).any(axis=1).mean()))
z = nz.reshape(256, 256, 256)
print("by octant (z,y,x halves): ", [(k, round(100 * z[(k >> 2) * 128:(k >> 2) * 128 + 128, ((k >> 1) & 1) * 128:((k >> 1) & 1) * 128 + 128, (k & 1) * 128:(k & 1) * 128 + 128].mean(), 2)) for k in range(8)])

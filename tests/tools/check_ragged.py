"""Volumes the chunk size does not divide against the reference build: python tests/tools/check_ragged.py [edge ...]
(a 1000^3 volume in 256^3 chunks has 27 regular chunks and 37 with an extent of 232, in 8 shape groups)"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch

from oracle import pyoracle
from sperr_amd.api import SperrHip
from sperr_amd.synth import turbulence_torch

eng = SperrHip()
impl = pyoracle.Ref() if pyoracle.have_ref() else pyoracle.Oracle()
shapes = [(int(a),) * 3 for a in sys.argv[1:]] or [(232, 256, 256), (232, 232, 232), (488, 256, 256)]
for shape in shapes:
    v = turbulence_torch(shape, "cuda", seed=7)
    c = eng.compress(v, (256, 256, 256), 2.0).clone()
    hv = v.cpu().numpy()
    want = impl.comp_3d(hv, (256, 256, 256), 1, 2.0, nthreads=64)
    same = bytes(c.cpu().numpy()) == want
    d = eng.decompress(c, True).cpu().numpy()
    r = impl.decomp_3d(want, True, nthreads=64)
    bad = np.argwhere(d.view(np.uint32) != r.view(np.uint32))
    print(shape, "container identical", same, "decode identical", bad.shape[0] == 0,
          "max err gpu %.4g ref %.4g" % (float(np.abs(d.astype(np.float64) - hv).max()),
                                         float(np.abs(r.astype(np.float64) - hv).max())), flush=True)
    if bad.shape[0]:
        print("  differing samples:", bad.shape[0], "first", bad[0], "last", bad[-1],
              "chunks (z,y,x)//256:", sorted({tuple(int(q) // 256 for q in b) for b in bad[:: max(1, bad.shape[0] // 2000)]})[:20])

"""Volumes the chunk size does not divide against the reference build (or the oracle):
python tests/tools/check_ragged.py [edge[:chunk[:bpp[:seed]]] ...]
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
for arg in sys.argv[1:] or ["488:256:2:7"]:
    f = arg.split(":")
    n, ch = int(f[0]), int(f[1]) if len(f) > 1 else 256
    bpp, seed = float(f[2]) if len(f) > 2 else 2.0, int(f[3]) if len(f) > 3 else 7
    shape = (n, n - n // 7, n - n // 3)    # (z, y, x): three different extents
    v = turbulence_torch(shape, "cuda", seed=seed)
    c = eng.compress(v, (ch, ch, ch), bpp).clone()
    hv = v.cpu().numpy()
    want = impl.comp_3d(hv, (ch, ch, ch), 1, bpp, nthreads=64)
    same = bytes(c.cpu().numpy()) == want
    d = eng.decompress(c, True).cpu().numpy()
    r = impl.decomp_3d(want, True, nthreads=64)
    bad = np.argwhere(d.view(np.uint32) != r.view(np.uint32))
    print(shape, "chunks", ch, "bpp", bpp, "container identical", same, "decode identical", bad.shape[0] == 0,
          "max err gpu %.4g ref %.4g" % (float(np.abs(d.astype(np.float64) - hv).max()),
                                         float(np.abs(r.astype(np.float64) - hv).max())), flush=True)
    if bad.shape[0]:
        print("  differing samples:", bad.shape[0], "first", bad[0], "last", bad[-1])

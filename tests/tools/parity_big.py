"""One-off byte-parity checks at sizes the test suite leaves out (the oracle needs minutes):
a 256^3 chunk in point-wise error mode, a 999 x 999 slice."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch

from oracle.pyoracle import Oracle
from sperr_amd.api import SperrHip
from sperr_amd.synth import turbulence

eng, o = SperrHip(), Oracle()
v = turbulence((256, 256, 256))
for tol in (1e-2, 1e-4):
    t = time.time()
    want = o.comp_3d(v, (256, 256, 256), 3, tol, nthreads=8)
    got = bytes(eng.compress(torch.from_numpy(v).cuda(), (256, 256, 256), tol, mode=3).cpu().numpy())
    dev = torch.from_numpy(np.frombuffer(want, dtype=np.uint8).copy()).cuda()
    same = np.array_equal(eng.decompress(dev, True).cpu().numpy().view(np.uint32), o.decomp_3d(want, True).view(np.uint32))
    print("256^3 PWE tol", tol, "bytes", len(want), "container identical:", got == want, "decode identical:", same,
          "(%.0f s)" % (time.time() - t))
img = turbulence((1, 999, 999))[0]
for mode, q in ((2, 90.0), (3, 1e-3)):
    want = o.comp_2d(img, mode, q, True)
    got = bytes(eng.compress_2d(torch.from_numpy(img).cuda(), q, mode=mode, header=True).cpu().numpy())
    dev = torch.from_numpy(np.frombuffer(want[10:], dtype=np.uint8).copy()).cuda()
    same = np.array_equal(eng.decompress_2d(dev, (999, 999), False).cpu().numpy().view(np.uint64),
                          o.decomp_2d(want[10:], (999, 999), False).view(np.uint64))
    print("999x999 mode", mode, "bytes", len(want), "stream identical:", got == want, "decode identical:", same)

# more chunks than one batch holds, with merged / ragged border chunks: 700 x 600 x 1300 (x y z) in
# 128^3 chunks -> 5 x 5 x 10 = 250 chunks of several shapes (the regular ones decode through the
# table kernels, the others through the serial walk).  The reference build (oracle/_ref) is the
# checker when present.
from oracle import pyoracle
ref = pyoracle.Ref() if pyoracle.have_ref() else o
from sperr_amd.synth import turbulence_torch
dv = turbulence_torch((1300, 600, 700), "cuda")      # (the numpy generator needs minutes at this size)
big = dv.cpu().numpy()
t = time.time()
want = ref.comp_3d(big, (128, 128, 128), 1, 2.0, nthreads=0 if ref is not o else 8)
t_ref = time.time() - t
got = eng.compress(dv, (128, 128, 128), 2.0)
same_c = bytes(got.cpu().numpy()) == want
print("700x600x1300, 250 chunks of 128^3 (ragged borders), 2 bpp: bytes", len(want), "container identical:", same_c,
      "(checker %.1f s)" % t_ref, flush=True)
torch.cuda.synchronize()
t = time.time()
back = eng.decompress(got, True)
torch.cuda.synchronize()
t_dec = time.time() - t
torch.cuda.synchronize()
t = time.time()
back = eng.decompress(got, True)
torch.cuda.synchronize()
print("   GPU decompress %.2f s (first call, builds the plans of 8 chunk shapes), %.2f s (second call), max err %.4f"
      % (t_dec, time.time() - t, float((back - dv).abs().max())), flush=True)
t = time.time()
part = ref.decomp_3d(want, True, nthreads=0 if ref is not o else 8)
print("   decode identical:", np.array_equal(back.cpu().numpy().view(np.uint32), part.view(np.uint32)),
      "(checker %.1f s)" % (time.time() - t), flush=True)

"""BASELINE configs[3]: a 999 x 999 float slice through sperr_comp_2d, PSNR = 90 dB (the reference's
test_data/999x999.float is not shipped; a synthetic slice of the same size stands in), plus the
fixed-rate and point-wise-error modes.  Timings with the data resident in HBM."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch

from sperr_amd.api import SperrHip
from sperr_amd.synth import turbulence

eng = SperrHip()
img_np = turbulence((1, 999, 999))[0]
img = torch.from_numpy(img_np).cuda()
rng = float(img_np.max() - img_np.min())
have_ref = False
try:
    from oracle import pyoracle
    ref = pyoracle.Ref() if pyoracle.have_ref() else pyoracle.Oracle()
    have_ref = True
except Exception:   # noqa: BLE001
    pass
for mode, q, name in [(2, 90.0, "PSNR 90 dB"), (1, 2.0, "2 bpp"), (3, 1e-4 * rng, "PWE 1e-4 of the range")]:
    s = eng.compress_2d(img, q, mode=mode)
    torch.cuda.synchronize()
    t0 = time.time()
    s = eng.compress_2d(img, q, mode=mode).clone()
    torch.cuda.synchronize()
    t1 = time.time()
    out = eng.decompress_2d(s, (999, 999), True)
    torch.cuda.synchronize()
    t2 = time.time()
    out = eng.decompress_2d(s, (999, 999), True)
    torch.cuda.synchronize()
    t3 = time.time()
    mse = float(((out.double() - img.double()) ** 2).mean())
    line = "%-22s %8d B (%.2f bpp)  GPU compress %7.1f ms  decompress %7.1f ms  PSNR %.2f dB  max err %.3g" % (
        name, s.numel(), s.numel() * 8 / img.numel(), (t1 - t0) * 1e3, (t3 - t2) * 1e3,
        10 * np.log10(rng * rng / mse), float((out.double() - img.double()).abs().max()))
    if have_ref:
        a = time.time()
        cs = ref.comp_2d(img_np, mode, q, False)
        b = time.time()
        ref.decomp_2d(cs, (999, 999), True)
        c = time.time()
        line += "  | CPU reference compress %.1f ms decompress %.1f ms, identical stream: %s" % (
            (b - a) * 1e3, (c - b) * 1e3, cs == bytes(s.cpu().numpy()))
    print(line)

"""Point-wise error mode on a 512^3 volume (8 chunks of 256^3): time, size and outlier share."""
import os
import struct
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

from sperr_amd.api import SperrHip
from sperr_amd.synth import turbulence_torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
eng = SperrHip()
vol = turbulence_torch((n, n, n), "cuda")
print("range", float(vol.min()), float(vol.max()))
# one output buffer for all calls (the worst-case bound of mode 3 is 32.5 bytes per value; allocating
# it inside the timed call would time hipMalloc)
out_buf = torch.empty(eng.max_compressed_size(vol.shape, (256, 256, 256), 1.0, 3), dtype=torch.uint8, device="cuda")
for tol in [float(t) for t in (sys.argv[2:] or ["1e-2", "1e-3", "1e-4"])]:
    s = eng.compress(vol, (256, 256, 256), tol, mode=3, out=out_buf)
    torch.cuda.synchronize()
    t0 = time.time()
    s = eng.compress(vol, (256, 256, 256), tol, mode=3, out=out_buf).clone()
    torch.cuda.synchronize()
    t1 = time.time()
    out = eng.decompress(s, True)
    torch.cuda.synchronize()
    del out   # (so that the timed call reuses this block instead of allocating a second one inside the timing)
    t2 = time.time()
    out = eng.decompress(s, True)
    torch.cuda.synchronize()
    t3 = time.time()
    err = float((out.double() - vol.double()).abs().max())
    b = bytes(s[:20 + 4 * 4096].cpu().numpy())
    nch = (n // 256) ** 3
    lens = struct.unpack_from("<%dI" % nch, b, 20 if nch > 1 else 14)
    # outlier share of the first chunk's stream
    off = (20 if nch > 1 else 14) + 4 * nch
    head = bytes(s[off:off + 26].cpu().numpy())
    tb = struct.unpack_from("<Q", head, 18)[0]
    sl = 17 + 9 + (tb + 7) // 8
    obits = 0
    if lens[0] > sl + 9:
        oh = bytes(s[off + sl:off + sl + 9].cpu().numpy())
        obits = struct.unpack_from("<Q", oh, 1)[0]
        onbp = oh[0]
    else:
        onbp = 0
    gb = vol.numel() * 4 / 1e9
    print("tol %g: %.1f MB (%.2f bpp), chunk0 speck %d B, outlier %d bits (%d planes); compress %.1f ms (%.1f GB/s), "
          "decompress %.1f ms (%.1f GB/s), max err %.3g" %
          (tol, s.numel() / 1e6, s.numel() * 8 / vol.numel(), sl, obits, onbp, (t1 - t0) * 1e3, gb / (t1 - t0),
           (t3 - t2) * 1e3, gb / (t3 - t2), err))
    if os.environ.get("PWE_PROFILE"):
        eng.profile(True)
        s2 = eng.compress(vol, (256, 256, 256), tol, mode=3, out=out_buf)
        eng.decompress(s2, True)
        torch.cuda.synchronize()
        rep = eng.profile_report()
        eng.profile(False)
        for k, (ms, cnt) in sorted(rep.items(), key=lambda kv: -kv[1][0])[:8]:
            print("   %-40s %8.2f ms %5d" % (k, ms, cnt))

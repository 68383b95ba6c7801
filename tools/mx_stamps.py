"""Per-region tick counters of k_lis_mx (chunk 0; `slice`: a 999 x 999 slice at PSNR 90 dB through the 2D
coder, needs SPERR_HIP_LIS_MX=2):  python tools/mx_stamps.py [edge | slice]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from sperr_amd.api import SperrHip
from sperr_amd.synth import turbulence, turbulence_torch

eng = SperrHip()
eng.lib.sperrhip_debug_lis_stamps.argtypes = [C.c_int, C.c_void_p]
if len(sys.argv) > 1 and sys.argv[1] == "slice":
    img = torch.from_numpy(turbulence((1, 999, 999))[0]).cuda()
    s = eng.compress_2d(img, 90.0, mode=2).clone()
    eng.lib.sperrhip_debug_lis_stamps(1, None)
    eng.decompress_2d(s, (999, 999), True)
else:
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 250
    vol = turbulence_torch((n, n, n), "cuda")
    s = eng.compress(vol, (n, n, n), 2.0).clone()
    eng.lib.sperrhip_debug_lis_stamps(1, None)
    eng.decompress(s, True)
torch.cuda.synchronize()
out = (C.c_ulonglong * 64)()
eng.lib.sperrhip_debug_lis_stamps(0, out)
n = max(out[0], 1)
print("regions %d; ticks per region: load + rows %d, look-back wait %d, on the chain %d, expansion %d" %
      (out[0], out[1] // n, out[2] // n, out[3] // n, out[4] // n))
print("sum over the regions (ticks): rows %d, wait %d, chain %d, expansion %d" % (out[1], out[2], out[3], out[4]))
for k in range(5, 27):
    if out[k]:
        print("  counter %d: %d  (%d per region)" % (k, out[k], out[k] // n))
print("chain parts (ticks): ring fills %d (%d fills), walk %d (%d calls): tight loop %d (%d hops, %d words), "
      "general-path hops %d, zero runs %d, rounds inside entered sets %d (%d ticks)" %
      (out[5], out[6], out[7], out[8], out[9], out[10], out[14], out[11], out[15], out[12], out[13]))
print("general-path entries by reason: length of 254 bits and more %d, column outside the list's two groups %d, split leaves the rows %d, "
      "no column but a length from the children's %d" % (out[16], out[17], out[18], out[19]))

print("per region (ticks): from the state words to the first walk %d, the walk's set-up %d, its end (publishing, staged items, records, state) %d; of the first: working the state out and putting it into LDS %d" % (out[22] // n, out[23] // n, out[24] // n, out[25] // n))

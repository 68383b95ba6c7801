"""Per-phase tick counters of k_lis_mixed (chunk 0 of a volume of one chunk; `slice`: a 999 x 999 slice
at PSNR 90 dB through the 2D coder):  python tools/mixed_stamps.py [edge | slice]"""
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
names = {0: "restart: classes", 1: "restart: rows", 7: "walk", 2: "walker waits for helpers", 3: "expand (level end)", 4: "compact", 5: "zero skip"}
tot = sum(out[i] for i in names)
for i, nm in names.items():
    print("%-14s %12d ticks  %5.1f%%" % (nm, out[i], 100 * out[i] / max(tot, 1)))
w = max(out[9], 1)
print("windows %d (%d pipeline restarts): ticks/window walk %d, waiting for the helpers %d; stream bits/window %d" %
      (out[9], out[8], out[7] // w, out[2] // w, out[20] // w))
if out[28]:
    print("helpers (%d wavefronts per plane sum): ticks/window expand %d, entry classes %d, rows %d; whole phase %d" %
          (out[28], out[25] // w, out[26] // w, out[27] // w, out[29] // w))
if out[45]:
    print("phase as the walker sees it: walk done after %d ticks, phase over after %d" % (out[44] // w, out[45] // w))
if out[43]:
    print("first helper: %d barriers per window, %d ticks per window inside them" % (out[43] // w, out[42] // w))
print("helper wavefronts, ticks of work per window:", [out[32 + i] // w for i in range(8)])
print("walk: %d of the significant list entries in the per-word loop, %d words, %d ticks there" % (out[23], out[24], out[30]))
print("walk: ticks entering list entries %d, inside the sets walked into %d (%d sets finished)" % (out[31], out[40], out[41]))
print("walk: %d significant list entries, %d child steps in %d entered sets, %d zero runs; %d skip rounds; %d expanded items"
      % (out[16], out[17], out[18], out[19], out[21], out[22]))
if out[16] + out[17]:
    print("walk ticks per hop/step: %.0f" % (out[7] / (out[16] + out[17] + out[19])))

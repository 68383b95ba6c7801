"""Per-phase tick counters of the table-driven LIS decoder (chunk 0), for kernel tuning."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from sperr_amd.api import SperrHip
from sperr_amd.synth import turbulence_torch

eng = SperrHip()
vol = turbulence_torch((512, 512, 512), "cuda")
s = eng.compress(vol, (256, 256, 256), 2.0).clone()
eng.lib.sperrhip_debug_lis_stamps.argtypes = [C.c_int, C.c_void_p]
eng.lib.sperrhip_debug_lis_stamps(1, None)
eng.decompress(s, True)
torch.cuda.synchronize()
out = (C.c_ulonglong * 64)()
eng.lib.sperrhip_debug_lis_stamps(0, out)
names = ["load", "tables", "hopS", "P1", "P2", "P3", "P4", "expand", "compact", "windows", "place"]
tot = sum(out[i] for i in range(9)) + out[10]
for i, n in enumerate(names):
    print(n, out[i], "%.1f%%" % (100 * out[i] / tot) if i != 9 else "")
print("K  windows  table_ticks  ticks/window  bits  bits/window")
for k in range(1, 16):
    w, t, b = out[16 + k], out[32 + k], out[48 + k]
    if w:
        print(k, w, t, t // w, b, b // w)

if out[56]:
    print("k_lis_l1 chunk 0: blocks", out[56], "ticks/block: tables+memo", out[57] // out[56],
          "look-back wait", out[58] // out[56], "marks+sweeps", out[59] // out[56])

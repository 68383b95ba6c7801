"""Tick counters of k_lis_hi (a chunk of the last decoded batch), for kernel tuning."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from sperr_amd.api import SperrHip
from sperr_amd.synth import turbulence_torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
eng = SperrHip()
vol = turbulence_torch((n, n, n), "cuda")
s = eng.compress(vol, (256, 256, 256), 2.0).clone()
eng.decompress(s, True)
eng.lib.sperrhip_debug_lis_stamps.argtypes = [C.c_int, C.c_void_p]
eng.lib.sperrhip_debug_lis_stamps(1, None)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
eng.decompress(s, True)
torch.cuda.synchronize()
print("decompress with stamps: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
out = (C.c_ulonglong * 64)()
eng.lib.sperrhip_debug_lis_stamps(0, out)
nb = max(1, out[0])
print("regions", out[0], "zero regions", out[9])
for i, name in enumerate(["tables", "wait", "chain", "expand"]):
    print(f"{name:8s} total {out[1 + i]:12d} ticks  per region {out[1 + i] // nb:8d}")
for i, name in enumerate(["decide", "publish", "hop", "P2", "P3", "P4"]):
    print(f"  chain/{name:8s} per region {out[10 + i] // nb:8d}")
print("on-chain table builds", out[5], "hop rebuilds", out[6], "serial hops", out[7], "list passes", out[8])
print("table phase per region: load+decide %d, class tables %d, hop tables %d, serve mask %d" % tuple(out[40 + i] // nb for i in range(4)))
print("expansion per region: %.1f rounds, %d items" % (out[44] / nb, out[45] // nb))
n0, n1 = max(1, out[48]), max(1, out[56])
print("k_lis_l0 blocks %d: tables+memo %d, look-back %d, entry walks %d, marks %d, tokens %d ticks per block" %
      ((out[48],) + tuple(out[49 + i] // n0 for i in range(5))))
print("k_lis_l1 blocks %d: tables+memo %d, look-back %d, after %d (entry walks %d, marks %d, sweep 1 %d, sweep 2 %d) ticks per block" %
      ((out[56],) + tuple(out[57 + i] // n1 for i in range(7))))

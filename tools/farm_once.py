#!/usr/bin/env python3
"""One farm compression (and decompression) of a pinned 1024^3 volume, three times, for a profiler to look at the last:
    python tools/farm_once.py [mode: 1 = rate 2 bpp | 3 = PWE 1e-3 of the range] [size]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sperr_amd.api import SperrHip
from sperr_amd.synth import turbulence_torch

mode = int(sys.argv[1]) if len(sys.argv) > 1 else 3
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
eng = SperrHip()
lib = eng.lib
vol = turbulence_torch((S, S, S), torch.device("cuda", 0))
q = 2.0 if mode == 1 else 1e-3 * float(vol.max() - vol.min())
hvol = torch.empty(vol.shape, dtype=torch.float32).pin_memory()
hvol.copy_(vol)
del vol
libc = C.CDLL(None)
libc.free.argtypes = [C.c_void_p]
for r in range(3):
    dst, n = C.c_void_p(None), C.c_size_t(0)
    t0 = time.perf_counter()
    rc = lib.sperrhip_comp_3d_farm(hvol.data_ptr(), 1, S, S, S, 256, 256, 256, mode, C.c_double(q), 0, None, 0, C.byref(dst), C.byref(n))
    t1 = time.perf_counter()
    assert rc == 0, rc
    print("compress %d: %.1f ms = %.1f GB/s, %d bytes" % (r, (t1 - t0) * 1e3, hvol.numel() * 4 / (t1 - t0) / 1e9, n.value), flush=True)
    libc.free(dst)

#!/usr/bin/env python3
"""The chunk farm in point-wise error mode (BASELINE config 5 in small: a pinned host volume, PWE +
outlier coder): workers per device swept.   python tools/farm_pwe.py [tol] [size]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from sperr_amd.api import SperrHip
from sperr_amd.synth import turbulence_torch

tol = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-3
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
eng = SperrHip()
lib = eng.lib
vol = turbulence_torch((S, S, S), torch.device("cuda", 0))
hvol = torch.empty(vol.shape, dtype=torch.float32).pin_memory()
hvol.copy_(vol)
hout = torch.empty(vol.shape, dtype=torch.float32).pin_memory()
del vol
nbytes = hvol.numel() * 4
libc = C.CDLL(None)
libc.free.argtypes = [C.c_void_p]


def comp():
    dst, n = C.c_void_p(None), C.c_size_t(0)
    t0 = time.perf_counter()
    rc = lib.sperrhip_comp_3d_farm(hvol.data_ptr(), 1, S, S, S, 256, 256, 256, 3, tol, 0, None, 0, C.byref(dst), C.byref(n))
    t1 = time.perf_counter()
    assert rc == 0, rc
    return dst, n.value, t1 - t0


def decomp(dst, n):
    x, y, z = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
    t0 = time.perf_counter()
    rc = lib.sperrhip_decomp_3d_into(dst, n, 1, 0, None, 0, hout.data_ptr(), nbytes, C.byref(x), C.byref(y), C.byref(z))
    t1 = time.perf_counter()
    assert rc == 0, rc
    return t1 - t0


print(f"# PWE tol {tol}, {S}^3 fp32 pinned, 256^3 chunks")
# (workers, chunks per item or 0 = the farm's own choice, helper threads: 1 / 0 / -1 = the farm's own choice)
SWEEP = ((0, 0, -1), (3, 0, 0), (3, 0, 1), (2, 0, 1), (3, 8, 0), (3, 6, 0), (4, 6, 0), (3, 4, 0))
for w, item, helpers in SWEEP:
    for k, v in (("SPERR_HIP_FARM_WORKERS", w), ("SPERR_HIP_FARM_DEC_WORKERS", w), ("SPERR_HIP_FARM_ITEM", item)):
        if v:
            os.environ[k] = str(v)
        else:
            os.environ.pop(k, None)
    if helpers >= 0:
        os.environ["SPERR_HIP_FARM_ASYNC"] = str(helpers)
    else:
        os.environ.pop("SPERR_HIP_FARM_ASYNC", None)
    tc, td = [], []
    for r in range(3):
        d, n, a = comp()
        b = decomp(d, n)
        libc.free(d)
        if r:
            tc.append(a)
            td.append(b)
    err = float((hout - hvol).abs().max())
    print(f"workers {w or 'auto'} item {item or 'auto'} helpers {helpers if helpers >= 0 else 'auto'}: compress {min(tc) * 1e3:7.1f} ms {nbytes / min(tc) / 1e9:5.1f} GB/s   "
          f"decompress {min(td) * 1e3:7.1f} ms {nbytes / min(td) / 1e9:5.1f} GB/s   bytes {n} max err {err:.3g}", flush=True)

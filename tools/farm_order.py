#!/usr/bin/env python3
"""Does what ran before matter to the farm's point-wise-error compression?  (bench.py runs it after the fixed-rate
host path and reports 140 - 160 ms where tools/farm_pwe.py, a fresh process, measures 106 - 126.)"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sperr_amd.api import SperrHip
from sperr_amd.synth import turbulence_torch

S = 1024
eng = SperrHip()
lib = eng.lib
vol = turbulence_torch((S, S, S), torch.device("cuda", 0))
tol = 1e-3 * float(vol.max() - vol.min())
hvol = torch.empty(vol.shape, dtype=torch.float32, pin_memory=True)
hvol.copy_(vol)
hout = torch.empty(vol.shape, dtype=torch.float32, pin_memory=True)
nbytes = hvol.numel() * 4
libc = C.CDLL(None)
libc.free.argtypes = [C.c_void_p]


def farm(mode, q, reps=3, src=None, out=None):
    tc, td = [], []
    src = src or hvol.data_ptr()
    for r in range(reps):
        dst, n = C.c_void_p(None), C.c_size_t(0)
        t0 = time.perf_counter()
        rc = lib.sperrhip_comp_3d_farm(src, 1, S, S, S, 256, 256, 256, mode, float(q), 0, None, 0, C.byref(dst), C.byref(n))
        t1 = time.perf_counter()
        assert rc == 0, rc
        x, y, z = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        rc = lib.sperrhip_decomp_3d_into(dst, n.value, 1, 0, None, 0, out or hout.data_ptr(), nbytes, C.byref(x), C.byref(y), C.byref(z))
        t2 = time.perf_counter()
        assert rc == 0, rc
        libc.free(dst)
        if r:
            tc.append(t1 - t0)
            td.append(t2 - t1)
    return min(tc) * 1e3, min(td) * 1e3


def dev_steps(n):
    cbuf = torch.empty(eng.max_compressed_size(vol.shape, (256,) * 3, 2.0), dtype=torch.uint8, device="cuda")
    out = torch.empty_like(vol)
    for _ in range(n):
        s = eng.compress(vol, (256, 256, 256), 2.0, out=cbuf)
        eng.decompress(s, True, out=out, shape_zyx=vol.shape)
    torch.cuda.synchronize()


print("fresh process:            PWE compress %.1f ms decompress %.1f ms" % farm(3, tol), flush=True)
dev_steps(3)
print("after 3 device steps:     PWE compress %.1f ms decompress %.1f ms" % farm(3, tol), flush=True)
print("  (rate mode:                 compress %.1f ms decompress %.1f ms)" % farm(1, 2.0), flush=True)
print("after the rate-mode farm: PWE compress %.1f ms decompress %.1f ms" % farm(3, tol), flush=True)
import numpy as np
pvol = hvol.numpy().copy()
pout = np.empty_like(pvol)
print("  (rate mode, pageable:       compress %.1f ms decompress %.1f ms)" % farm(1, 2.0, 3, pvol.ctypes.data, pout.ctypes.data), flush=True)
print("after the pageable farm:  PWE compress %.1f ms decompress %.1f ms" % farm(3, tol), flush=True)
print("again:                    PWE compress %.1f ms decompress %.1f ms" % farm(3, tol), flush=True)
lib.sperrhip_release()
print("after sperrhip_release(): PWE compress %.1f ms decompress %.1f ms" % farm(3, tol), flush=True)

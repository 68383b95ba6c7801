#!/usr/bin/env python3
"""Sweep of the chunk farm's knobs on one host-resident 1024^3 volume (GPU box):
   python tools/farm_tune.py [pinned|pageable]  ->  one line per setting."""
import ctypes as C
import itertools
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from sperr_amd.api import SperrHip
from sperr_amd.synth import turbulence_torch

kind = sys.argv[1] if len(sys.argv) > 1 else "pinned"
S = int(os.environ.get("TUNE_SIZE", "1024"))
eng = SperrHip()
lib = eng.lib
vol = turbulence_torch((S, S, S), torch.device("cuda", 0))
nbytes = vol.numel() * 4
if kind == "pinned":
    hvol = torch.empty(vol.shape, dtype=torch.float32).pin_memory()
    hvol.copy_(vol)
    hout = torch.empty(vol.shape, dtype=torch.float32).pin_memory()
    src, out = hvol.data_ptr(), hout.data_ptr()
else:
    hvol = vol.cpu().numpy()
    hout = np.empty_like(hvol)
    src, out = hvol.ctypes.data, hout.ctypes.data
del vol
libc = C.CDLL(None)
libc.free.argtypes = [C.c_void_p]
chunks = (256, 256, 256)


def comp():
    dst, n = C.c_void_p(None), C.c_size_t(0)
    t0 = time.perf_counter()
    rc = lib.sperrhip_comp_3d_farm(src, 1, S, S, S, *chunks, 1, 2.0, 0, None, 0, C.byref(dst), C.byref(n))
    t1 = time.perf_counter()
    assert rc == 0
    return dst, n.value, t1 - t0


def decomp(dst, n):
    x, y, z = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
    t0 = time.perf_counter()
    rc = lib.sperrhip_decomp_3d_into(dst, n, 1, 0, None, 0, out, nbytes, C.byref(x), C.byref(y), C.byref(z))
    t1 = time.perf_counter()
    assert rc == 0
    return t1 - t0


def setenv(**kv):
    for k, v in kv.items():
        os.environ[k] = str(v)


dst0, n0, _ = comp()
print(f"# {kind} {S}^3; GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')}")
print("# compress: workers item helpers copy -> ms GB/s")
CW = tuple(int(v) for v in os.environ.get("TUNE_CW", "2,3,4").split(","))
CI = tuple(int(v) for v in os.environ.get("TUNE_CI", "4,8,11,16").split(","))
DW = tuple(int(v) for v in os.environ.get("TUNE_DW", "2,3,4").split(","))
DI = tuple(int(v) for v in os.environ.get("TUNE_DI", "8,16,22,32").split(","))
for w, item, hl, cp in itertools.product(CW, CI, (4, 8) if kind != "pinned" else (4,),
                                         ("3d", "stage") if kind == "pinned" else ("stage",)):
    setenv(SPERR_HIP_FARM_WORKERS=w, SPERR_HIP_FARM_ITEM=item, SPERR_HIP_FARM_HELPERS=hl, SPERR_HIP_PINNED_COPY=cp)
    ts = []
    for r in range(3):
        d, n, t = comp()
        libc.free(d)
        if r:
            ts.append(t)
    print(f"C w={w} item={item} helpers={hl} copy={cp}: {min(ts) * 1e3:.1f} ms {nbytes / min(ts) / 1e9:.1f} GB/s", flush=True)
print("# decompress: workers item helpers copy -> ms GB/s")
for w, item, hl, cp in itertools.product(DW, DI, (4, 8) if kind != "pinned" else (4,),
                                         ("3d", "stage") if kind == "pinned" else ("stage",)):
    setenv(SPERR_HIP_FARM_DEC_WORKERS=w, SPERR_HIP_FARM_ITEM=item, SPERR_HIP_FARM_HELPERS=hl, SPERR_HIP_PINNED_COPY=cp)
    ts = []
    for r in range(3):
        t = decomp(dst0, n0)
        if r:
            ts.append(t)
    print(f"D w={w} item={item} helpers={hl} copy={cp}: {min(ts) * 1e3:.1f} ms {nbytes / min(ts) / 1e9:.1f} GB/s", flush=True)

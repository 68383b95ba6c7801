"""Slices decoded (and encoded) from several host threads at once: every call takes an engine of the
device's pool (SPERR_HIP_ENGINES_PER_DEVICE, default 4), and a slice keeps one workgroup busy, so the
calls run side by side:  python tools/slice_threads.py [threads ...]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from sperr_amd.api import SperrHip
from sperr_amd.synth import turbulence

counts = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
eng = SperrHip()
imgs = [torch.from_numpy(turbulence((1, 999, 999), seed=s)[0]).cuda() for s in range(max(counts))]
streams = [eng.compress_2d(im, 90.0, mode=2).clone() for im in imgs]
ref = [eng.decompress_2d(s, (999, 999), True).clone() for s in streams]
torch.cuda.synchronize()
for n in counts:
    for what in ("decompress", "compress"):
        reps = 4
        ok = [True] * n

        def work(i):   # (a stream per thread: calls on one stream would queue behind each other)
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for _ in range(reps):
                    if what == "decompress":
                        out = eng.decompress_2d(streams[i], (999, 999), True)
                        ok[i] = ok[i] and bool(torch.equal(out, ref[i]))
                    else:
                        s = eng.compress_2d(imgs[i], 90.0, mode=2)
                        ok[i] = ok[i] and bool(torch.equal(s, streams[i]))
                st.synchronize()

        ts = [threading.Thread(target=work, args=(i,)) for i in range(n)]
        torch.cuda.synchronize()
        t0 = time.time()
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        torch.cuda.synchronize()
        dt = time.time() - t0
        print("%d thread(s): %s %.1f slices/s (%.1f ms per slice and thread), results identical: %s"
              % (n, what, n * reps / dt, dt / reps * 1e3, all(ok)), flush=True)


# ---- the host entry points (sperr_comp_2d / sperr_decomp_2d: host buffers in and out, malloc'd results)
import ctypes as C

import numpy as np

lib = eng.lib
lib.sperr_comp_2d.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.c_int, C.c_double, C.c_int,
                              C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
lib.sperr_decomp_2d.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_size_t, C.c_size_t, C.POINTER(C.c_void_p)]
libc = C.CDLL(None)
libc.free.argtypes = [C.c_void_p]
himgs = [im.cpu().numpy() for im in imgs]
hstreams = [bytes(s.cpu().numpy()) for s in streams]
href = [r.cpu().numpy() for r in ref]
for n in counts:
    for what in ("decompress", "compress"):
        reps = 4
        ok = [True] * n

        def hwork(i):
            for _ in range(reps):
                dst = C.c_void_p(None)
                if what == "decompress":
                    rc = lib.sperr_decomp_2d(hstreams[i], len(hstreams[i]), 1, 999, 999, C.byref(dst))
                    out = np.ctypeslib.as_array(C.cast(dst, C.POINTER(C.c_float)), shape=(999, 999)) if rc == 0 else None
                    ok[i] = ok[i] and rc == 0 and np.array_equal(out, href[i])
                else:
                    ln = C.c_size_t(0)
                    rc = lib.sperr_comp_2d(himgs[i].ctypes.data, 1, 999, 999, 2, 90.0, 0, C.byref(dst), C.byref(ln))
                    ok[i] = ok[i] and rc == 0 and C.string_at(dst.value, ln.value) == hstreams[i]
                if dst.value:
                    libc.free(dst)

        ts = [threading.Thread(target=hwork, args=(i,)) for i in range(n)]
        t0 = time.time()
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        dt = time.time() - t0
        print("%d thread(s): host %s %.1f slices/s (%.1f ms per slice and thread), results identical: %s"
              % (n, what, n * reps / dt, dt / reps * 1e3, all(ok)), flush=True)

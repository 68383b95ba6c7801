#!/usr/bin/env python3
"""BASELINE config 5 at size on ONE GPU: a fp32 volume far larger than the device's working set, in pinned host
memory, point-wise error mode + outlier coder, 256^3 chunks, streamed through the chunk farm
(sperrhip_comp_3d_farm / sperrhip_decomp_3d_into -- what sperr_comp_3d / sperr_decomp_3d run on).  The reference's
counterpart: /root/reference/src/SPERR3D_OMP_C.cpp:61-141 (chunk loop), :163-234 (header + length table).

  python tools/config5_big.py [--edge 2048 | --edge auto] [--tol-rel 1e-3] [--sample 16] [--full]

The volume is generated slab by slab ON THE GPU straight into pinned memory (it never exists on the device as a
whole).  Reported: sustained GB/s each way (H2D/D2H inside), peak RSS, pinned bytes (the two volumes + the farm's
staging), the device buffers of the farm and the largest engine arena, the tolerance check over every value
(slab-wise on the GPU), header + length-table consistency for all chunks, and byte parity of `--sample` chunks
against the oracle (each sampled chunk compressed alone by the CPU oracle: its stream must be the bytes the big
container holds for that chunk).  --full: the whole container against the oracle's (RAM and minutes permitting).
"""
import argparse
import ctypes as C
import os
import resource
import struct
import sys
import time
from concurrent.futures import ThreadPoolExecutor

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from sperr_amd.api import SperrHip
from sperr_amd.synth import turbulence_torch


def mem_available():
    """What this job may take: /proc/meminfo's MemAvailable is the HOST's, a pod's share can be far smaller
    (cgroup v2 memory.max / v1 memory.limit_in_bytes, minus what is in use); the smaller of the two."""
    avail = 0
    for line in open("/proc/meminfo"):
        if line.startswith("MemAvailable:"):
            avail = int(line.split()[1]) * 1024
    rel = "/"
    try:
        for line in open("/proc/self/cgroup"):
            f = line.strip().split(":")
            if f[1] in ("", "memory"):
                rel = f[2]
    except OSError:
        pass
    for base, lim, cur in (("/sys/fs/cgroup", "memory.max", "memory.current"),
                           ("/sys/fs/cgroup/memory", "memory.limit_in_bytes", "memory.usage_in_bytes")):
        for d in (base + rel, base):
            try:
                v = open(os.path.join(d, lim)).read().strip()
                if v != "max" and int(v) < (1 << 60):
                    used = int(open(os.path.join(d, cur)).read().strip())
                    avail = min(avail, max(0, int(v) - used))
            except (OSError, ValueError):
                pass
    return avail


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--edge", default="auto")
    ap.add_argument("--tol-rel", type=float, default=1e-3)
    ap.add_argument("--sample", type=int, default=16)
    ap.add_argument("--full", action="store_true")
    ap.add_argument("--chunk", type=int, default=256)
    ap.add_argument("--max-edge", type=int, default=2048, help="largest edge `--edge auto` may choose")
    args = ap.parse_args()
    avail = mem_available()
    # Two pinned volumes + the container + the farm's staging: 2.4 x the volume.  Pinned memory cannot be
    # reclaimed, and a box whose pod runs out of memory is LOST (round 4 lost one to `auto` choosing 4096^3
    # from the host's MemAvailable): never more than 35 % of what is available, `auto` never above
    # --max-edge, and an explicit --edge is refused when it does not fit either.
    fits = lambda e: 2.4 * 4 * e ** 3 < 0.35 * avail
    if args.edge == "auto":
        edge = next((e for e in (4096, 3072, 2048, 1536, 1024) if e <= args.max_edge and fits(e)), 0)
    else:
        edge = int(args.edge)
    if edge == 0 or not fits(edge):
        print(f"# refused: {edge or args.max_edge}^3 needs {2.4 * 4 * (edge or 1024) ** 3 / 2**30:.0f} GiB of host memory, "
              f"35 % of what this job may take is {0.35 * avail / 2**30:.0f} GiB")
        sys.exit(3)
    S, Cn = edge, args.chunk
    nvals = S ** 3
    nbytes = 4 * nvals
    nch = (S // Cn) ** 3
    print(f"# config 5 at size: {S}^3 fp32 = {nbytes / 2**30:.1f} GiB in pinned host memory, {nch} chunks of {Cn}^3, "
          f"PWE mode + outlier coder, one GPU; host MemAvailable {avail / 2**30:.0f} GiB, {os.cpu_count()} threads",
          flush=True)
    eng = SperrHip()
    lib = eng.lib
    lib.sperrhip_debug_counter.restype = C.c_ulonglong
    lib.sperrhip_debug_counter.argtypes = [C.c_int]
    dev = torch.device("cuda", 0)
    t0 = time.perf_counter()
    hvol = torch.empty((S, S, S), dtype=torch.float32, pin_memory=True)   # (pinned from the start: no pageable twin)
    hout = torch.empty((S, S, S), dtype=torch.float32, pin_memory=True)
    print(f"pinned 2 x {nbytes / 2**30:.1f} GiB in {time.perf_counter() - t0:.1f} s", flush=True)
    t0 = time.perf_counter()
    slab = 64
    lo, hi = float("inf"), float("-inf")
    for z0 in range(0, S, slab):
        part = turbulence_torch((min(slab, S - z0), S, S), dev, seed=5, origin_z=z0, period=float(Cn))
        lo, hi = min(lo, float(part.min())), max(hi, float(part.max()))
        hvol[z0:z0 + part.shape[0]].copy_(part)
        if z0 % 512 == 0:
            print(f"  generated planes {z0}..", flush=True)
    del part
    torch.cuda.synchronize()
    tol = args.tol_rel * (hi - lo)
    print(f"generated in {time.perf_counter() - t0:.1f} s; range [{lo:.4f}, {hi:.4f}], tolerance {tol:.6g}", flush=True)

    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    best = {}
    keep = None
    for rep in range(2):
        dst, n = C.c_void_p(None), C.c_size_t(0)
        t0 = time.perf_counter()
        rc = lib.sperrhip_comp_3d_farm(hvol.data_ptr(), 1, S, S, S, Cn, Cn, Cn, 3, float(tol), 0, None, 0,
                                       C.byref(dst), C.byref(n))
        tc = time.perf_counter() - t0
        assert rc == 0, f"sperrhip_comp_3d_farm returned {rc}"
        x, y, z = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        t0 = time.perf_counter()
        rc = lib.sperrhip_decomp_3d_into(dst, n.value, 1, 0, None, 0, hout.data_ptr(), nbytes, C.byref(x),
                                         C.byref(y), C.byref(z))
        td = time.perf_counter() - t0
        assert rc == 0, f"sperrhip_decomp_3d_into returned {rc}"
        print(f"pass {rep}: compress {tc:7.3f} s = {nbytes / tc / 1e9:6.2f} GB/s   decompress {td:7.3f} s = "
              f"{nbytes / td / 1e9:6.2f} GB/s   container {n.value} B = {n.value * 8 / nvals:.3f} bpp", flush=True)
        best["c"] = min(best.get("c", 1e9), tc)
        best["d"] = min(best.get("d", 1e9), td)
        if keep is not None:
            libc.free(keep[0])
        keep = (dst, n.value)
    dst, n = keep
    cont = np.ctypeslib.as_array(C.cast(dst, C.POINTER(C.c_uint8)), shape=(n,))
    print(f"sustained: compress {nbytes / best['c'] / 1e9:.2f} GB/s, decompress {nbytes / best['d'] / 1e9:.2f} GB/s "
          "(best of 2 passes; the first also allocates the staging buffers)")
    print(f"peak RSS {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2**20:.2f} GiB; pinned: volumes "
          f"{2 * nbytes / 2**30:.2f} GiB + farm staging {lib.sperrhip_debug_counter(4) / 2**30:.2f} GiB; device: farm buffers "
          f"{lib.sperrhip_debug_counter(5) / 2**30:.2f} GiB + largest engine arena {lib.sperrhip_debug_counter(3) / 2**30:.2f} GiB")

    # ---- tolerance over every value, slab-wise on the GPU
    worst = 0.0
    for z0 in range(0, S, slab):
        a = hvol[z0:z0 + slab].to(dev, non_blocking=True).double()
        b = hout[z0:z0 + slab].to(dev, non_blocking=True).double()
        worst = max(worst, float((a - b).abs().max()))
    ok_tol = worst <= tol + 6e-8 * max(abs(lo), abs(hi))    # (fp32 output: half an ulp of the largest value on top)
    print(f"max |error| {worst:.6g} <= tolerance {tol:.6g}: {ok_tol}")

    # ---- header + length table (src/SPERR3D_OMP_C.cpp:163-234)
    hdr = 20 + 4 * nch
    flags = cont[1]
    v3 = struct.unpack_from("<3I", cont, 2)
    c3 = struct.unpack_from("<3H", cont, 14)
    lens = np.frombuffer(cont[20:hdr].tobytes(), dtype=np.uint32).astype(np.int64)
    ok_hdr = (cont[0] == 0 and flags == (0x40 | 0x20 | 0x10) and v3 == (S, S, S) and c3 == (Cn, Cn, Cn)
              and hdr + int(lens.sum()) == n and bool((lens > 26).all()))
    print(f"header: version {cont[0]}, flags {flags:#x}, volume {v3}, chunks {c3}; length table of {nch} entries, "
          f"sum + header == container length: {hdr + int(lens.sum()) == n}; consistent: {ok_hdr}")
    offs = hdr + np.concatenate([[0], np.cumsum(lens)[:-1]])

    # ---- byte parity of sampled chunks against the oracle (each chunk compressed alone on the CPU)
    from oracle import pyoracle
    impl = pyoracle.Ref() if pyoracle.have_ref() else pyoracle.Oracle()
    kind = "reference build" if pyoracle.have_ref() else "C restatement"
    per = S // Cn
    pick = sorted(set(int(round(i * (nch - 1) / max(1, args.sample - 1))) for i in range(args.sample)))
    hv = hvol.numpy()

    def one(g):
        cz, cy, cx = g // (per * per), (g // per) % per, g % per
        blk = np.ascontiguousarray(hv[cz * Cn:(cz + 1) * Cn, cy * Cn:(cy + 1) * Cn, cx * Cn:(cx + 1) * Cn])
        want = impl.comp_3d(blk, (Cn, Cn, Cn), 3, float(tol), nthreads=1)
        mine = cont[offs[g]:offs[g] + lens[g]].tobytes()
        return g, want[18:] == mine, len(want) - 18, int(lens[g])

    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=min(16, len(pick))) as ex:
        res = list(ex.map(one, pick))
    bad = [r for r in res if not r[1]]
    print(f"oracle ({kind}) on {len(pick)} sampled chunks {pick} in {time.perf_counter() - t0:.1f} s: "
          f"{len(pick) - len(bad)} byte-identical, {len(bad)} differ {bad}")
    ok_full = None
    if args.full:
        t0 = time.perf_counter()
        threads = max(1, min(os.cpu_count() or 1, nch, int(mem_available() / (3 * 2**30))))
        want = impl.comp_3d(hv, (Cn, Cn, Cn), 3, float(tol), nthreads=threads)
        ok_full = (len(want) == n) and (np.frombuffer(want, dtype=np.uint8) == cont).all()
        print(f"whole container against the oracle ({kind}, {threads} threads, {time.perf_counter() - t0:.1f} s: "
              f"{nbytes / (time.perf_counter() - t0) / 1e9:.2f} GB/s): identical {bool(ok_full)}")
    libc.free(dst)
    good = ok_tol and ok_hdr and not bad and ok_full is not False
    print("RESULT", "ok" if good else "FAILED")
    sys.exit(0 if good else 1)


if __name__ == "__main__":
    main()

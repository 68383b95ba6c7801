#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on one node: SPERR 3D fixed-rate compress + decompress
throughput (GB/s of uncompressed fp32 bytes) on 256^3 chunks at BPP = 2.0.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the hot path over one synthetic volume already resident in HBM:
compress the volume into a SPERR container (device memory) and decompress that container back
into a device volume.  Every rank owns its own volume of `--size`^3 fp32 samples cut into
256^3 chunks (independent chunks, no data-path collective: weak scaling).  `value` is the
whole-job rate: N * volume bytes / max-over-ranks(time per step).

Besides the contract fields the JSON line carries
  roofline      the dominant kernel's algorithmic bytes / its measured time vs the HBM peak
  cpu_baseline  the CPU oracle (the real reference build when oracle/_ref is present, else this
                repo's C restatement) timed on the same volume on this host (all threads: the
                reference's OpenMP loop runs one chunk per thread, so at most 64 threads work) and
                on one 256^3 chunk with one thread
  host_path     ONE volume in (pinned) host memory through the library's chunk farm
                (sperrhip_comp_3d_farm / sperrhip_decomp_3d_into: what sperr_comp_3d runs on) over
                the N devices of this launch, transfers included: the H2D/D2H-inclusive rate of
                SURVEY 8(d) at N = 1, and the strong-scaling figure (a fixed 64-chunk volume dealt
                to N GPUs) at N > 1.  Driven by rank 0; never used as `value`.
  other_modes   the bench volume in point-wise error mode (tolerance 1e-3 of the range) and one
                999 x 999 slice at PSNR 90 dB through the 2D coder.  Rank 0; never used as `value`.
  ragged_volume a 1000^3 volume in the same 256^3 chunks: the chunk size does not divide it, so most
                chunks have extents that are not powers of two (decoded by k_lis_mx).  Rank 0;
                never used as `value`.
"""
import argparse
import json
import os
import sys
import time

# the decoder runs the shape groups of a volume side by side on up to 8 streams: ask the runtime for
# as many hardware queues (its default is 4; must be set before the first HIP call of the process)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
ALGO_BYTES_PER_VALUE = 4.25    # 4 B fp32 + BPP/8 B stream per value


def run_host_path(eng, vol, chunks, bpp, devices, reps, dev_stream, dev_out, mode=1, pageable=True):
    """One volume in pinned host memory through the chunk farm on `devices`; transfers included.
    mode 1: `bpp` bits per value; mode 3: `bpp` is the point-wise tolerance (BASELINE config 5's
    combination: pinned host volume + PWE + outlier coder + 256^3 chunks, work items streaming
    through the device)."""
    import ctypes as C
    import numpy as np
    import torch
    lib = eng.lib
    dz, dy, dx = vol.shape
    nbytes = vol.numel() * 4
    hvol = torch.empty(vol.shape, dtype=torch.float32).pin_memory()
    hvol.copy_(vol)
    hout = torch.empty(vol.shape, dtype=torch.float32).pin_memory()
    arr = (C.c_int * len(devices))(*devices)
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]

    def comp(ptr):
        dst, n = C.c_void_p(None), C.c_size_t(0)
        t0 = time.perf_counter()
        rc = lib.sperrhip_comp_3d_farm(ptr, 1, dx, dy, dz, *chunks, mode, float(bpp), 0, arr, len(devices),
                                       C.byref(dst), C.byref(n))
        t1 = time.perf_counter()
        assert rc == 0, f"sperrhip_comp_3d_farm returned {rc}"
        return dst, n.value, t1 - t0

    def decomp(dst, n, out_ptr):
        x, y, z = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        t0 = time.perf_counter()
        rc = lib.sperrhip_decomp_3d_into(dst, n, 1, 0, arr, len(devices), out_ptr, nbytes, C.byref(x),
                                         C.byref(y), C.byref(z))
        t1 = time.perf_counter()
        assert rc == 0, f"sperrhip_decomp_3d_into returned {rc}"
        return t1 - t0

    from sperr_amd.api import host_throttle
    tc, td = [], []
    same = None
    thr0 = host_throttle(lib)
    for it in range(reps + 1):          # the first pass allocates the staging buffers: untimed
        dst, n, a = comp(hvol.data_ptr())
        b = decomp(dst, n, hout.data_ptr())
        if it == 0:
            if dev_stream is not None:
                got = np.ctypeslib.as_array(C.cast(dst, C.POINTER(C.c_uint8)), shape=(n,))
                same = bool(n == dev_stream.numel() and
                            torch.equal(torch.from_numpy(got.copy()), dev_stream.cpu()))
        else:
            tc.append(a)
            td.append(b)
        libc.free(dst)
    thr1 = host_throttle(lib)
    c, d = min(tc), min(td)
    what = f"BPP {bpp}" if mode == 1 else f"point-wise error mode, tolerance {bpp:.4g}, outlier coder"
    res = {
        "what": f"one {dx}x{dy}x{dz} fp32 volume in pinned host memory, chunks {chunks}, {what}, farmed over "
                f"{len(devices)} device(s) by the library (sperrhip_comp_3d_farm / sperrhip_decomp_3d_into); "
                "H2D of the volume and D2H of the container (and back) inside the timing; best of "
                f"{reps} after one untimed pass",
        "scaling": "strong", "n_gpus": len(devices),
        "compress_GBps": round(nbytes / c / 1e9, 3), "decompress_GBps": round(nbytes / d / 1e9, 3),
        "round_trip_GBps": round(nbytes / (c + d) / 1e9, 3),
        "compress_ms": round(c * 1e3, 2), "decompress_ms": round(d * 1e3, 2),
        "container_bytes": int(n),
        # CFS throttling of this process's cgroup over the reps + 1 passes above (cpu.stat deltas): a host
        # path that misses its rate on a box whose quota was exhausted says so here
        "cfs_throttle": (None if thr0 is None or thr1 is None else
                         {"nr_throttled": thr1[0] - thr0[0], "throttled_ms": round((thr1[1] - thr0[1]) / 1e3, 2)}),
    }
    if mode == 3:
        err = float((hout.double() - hvol.double()).abs().max().item())
        res["max_abs_err"] = err
        # (the decoder narrows to fp32: half an fp32 ulp of the largest value comes on top)
        res["within_tolerance"] = bool(err <= float(bpp) + 6e-8 * float(hvol.abs().max()))
    if dev_stream is not None:
        res["container_identical_to_device_path"] = same
    if pageable:
        # the same from pageable memory (numpy): rows are staged by helper threads
        pvol = hvol.numpy().copy()
        dst, n, _ = comp(pvol.ctypes.data)
        libc.free(dst)
        dst, n, pc = comp(pvol.ctypes.data)
        pout = np.empty_like(pvol)
        decomp(dst, n, pout.ctypes.data)
        pd = decomp(dst, n, pout.ctypes.data)
        libc.free(dst)
        res["pageable_compress_GBps"] = round(nbytes / pc / 1e9, 3)
        res["pageable_decompress_GBps"] = round(nbytes / pd / 1e9, 3)
        if dev_out is not None:
            res["decoded_identical_to_device_path"] = bool(torch.equal(hout, dev_out.cpu()) and
                                                           torch.equal(hout, torch.from_numpy(pout)))
    elif dev_out is not None:
        res["decoded_identical_to_device_path"] = bool(torch.equal(hout, dev_out.cpu()))
    return res


def compact_line(line):
    """The bench line without its prose: "what" strings dropped, other long strings cut, bookkeeping that a
    reader of the driver's record does not need (`traffic_from` beyond its file, `farm_threads.host`) folded."""
    def strip(o, depth=0):
        if isinstance(o, dict):
            out = {}
            for k, v in o.items():
                if k == "what":
                    continue
                out[k] = v if k == "config" else strip(v, depth + 1)   # (config.workload names the workload: whole)
            return out
        if isinstance(o, list):
            return [strip(v, depth + 1) for v in o]
        if isinstance(o, str) and len(o) > 120 and depth > 0:
            return o[:117] + "..."
        return o
    c = strip(line)
    rf = c.get("roofline") or {}
    if isinstance(rf.get("traffic_from"), dict):
        rf["traffic_from"] = rf["traffic_from"].get("file") or rf["traffic_from"].get("why", "")[:60]
    ft = c.get("farm_threads")
    if isinstance(ft, dict):
        c["farm_threads"] = {k: (v.get("threads_total") if isinstance(v, dict) and "threads_total" in v else v)
                             for k, v in ft.items() if k != "host"}
    return c


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=1024, help="volume edge per GPU")
    ap.add_argument("--chunk", type=int, default=256)
    ap.add_argument("--bpp", type=float, default=2.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=1024, help="edge of the CPU baseline sample")
    ap.add_argument("--profile-out", default="", help="write the per-kernel event table here")
    ap.add_argument("--no-host-path", action="store_true", help="skip the host-resident farm run")
    ap.add_argument("--host-reps", type=int, default=3)
    ap.add_argument("--no-small-batch", action="store_true", help="skip the 8-chunk batch")
    ap.add_argument("--no-ragged", action="store_true", help="skip the volume the chunk size does not divide")
    ap.add_argument("--ragged-size", type=int, default=1000)
    ap.add_argument("--no-other-modes", action="store_true", help="skip the point-wise error run and the 2D slice")
    ap.add_argument("--verbose", action="store_true", help="the JSON line with its descriptions (about 12 KB)")
    ap.add_argument("--full-out", default="", help="write the verbose record to this file as well")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    host_pg = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl")
        # a host-side group: ranks that wait for rank 0's farm run must not spin on their GPUs
        host_pg = dist.new_group(backend="gloo")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from sperr_amd.api import SperrHip
    from sperr_amd.synth import turbulence_torch

    eng = SperrHip()   # raises if libsperr_hip.so is missing: there is no fallback path
    S, C = args.size, args.chunk
    vol = turbulence_torch((S, S, S), dev, seed=42 + rank)
    nbytes = vol.numel() * 4
    chunks = (C, C, C)
    cap = eng.max_compressed_size(vol.shape, chunks, args.bpp)
    cbuf = torch.empty(cap, dtype=torch.uint8, device=dev)
    out = torch.empty_like(vol)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step(timers=None):
        if timers is not None:
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record()
        stream = eng.compress(vol, chunks, args.bpp, out=cbuf)
        if timers is not None:
            e1.record()
        eng.decompress(stream, output_float=True, out=out, shape_zyx=vol.shape)
        if timers is not None:
            e2.record()
            timers.append((e0, e1, e2))
        return stream

    for _ in range(args.warmup):
        step()
    # one untimed step with every kernel bracketed by HIP events finds the dominant kernel and the
    # per-kernel table; in the timed region only the dominant kernel is bracketed (two event
    # records per launch of EVERY kernel cost about 6 % of a step)
    eng.profile(True)
    step()
    torch.cuda.synchronize()
    eng.profile(False)
    prof_all = eng.profile_report(with_sum=True)   # {kernel: (busy ms, launches, sum of durations)}
    # dominant = largest SUM of launch durations (launches x average duration, what a kernel trace reports)
    top_name = max(prof_all.items(), key=lambda kv: kv[1][2])[0]
    eng.profile(True, only=top_name)
    timers = []
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        stream = step(timers)
    barrier()
    t1 = time.perf_counter()
    eng.profile(False)
    prof = eng.profile_report(with_sum=True)

    dt = (t1 - t0) / args.steps
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt_max = float(tmax.item())
    tc = sum(a.elapsed_time(b) for a, b, _ in timers) / len(timers) / 1e3
    td = sum(b.elapsed_time(c) for _, b, c in timers) / len(timers) / 1e3

    # correctness on the timed data: exact container length, bounded round-trip error
    nchunks = (S // C) ** 3
    exp_len = (20 if nchunks > 1 else 14) + 4 * nchunks + nchunks * (17 + 9 + int(args.bpp * C ** 3) // 8)
    err = float((out.double() - vol.double()).abs().max().item())
    ok_len = int(stream.numel()) == exp_len

    # ---- one host-resident volume through the library's farm over all N devices (rank 0) ------
    host_path = host_path_pwe = None
    if not args.no_host_path:
        if rank == 0:
            # (a launcher may show every rank only its own GPU: then the farm has that one)
            ndev = max(1, min(world, torch.cuda.device_count()))
            try:
                host_path = run_host_path(eng, vol, chunks, args.bpp, list(range(ndev)), args.host_reps, stream, out)
            except Exception as ex:   # never lose the line of the weak-scaling run over this
                host_path = {"error": f"{type(ex).__name__}: {ex}", "n_gpus": ndev}
            # BASELINE configs[4]'s combination on the devices of this launch: the same pinned volume in
            # point-wise error mode (tolerance 1e-3 of the range), 256^3 chunks, streamed item by item
            try:
                tol5 = 1e-3 * float(vol.max() - vol.min())
                host_path_pwe = run_host_path(eng, vol, chunks, tol5, list(range(ndev)), max(1, args.host_reps - 1),
                                              None, None, mode=3, pageable=False)
            except Exception as ex:
                host_path_pwe = {"error": f"{type(ex).__name__}: {ex}", "n_gpus": ndev}
        if world > 1:
            dist.barrier(group=host_pg)

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # ---- device workspace of the call: the engine's arena holds every chunk of the batch in flight
    workspace = None
    try:
        import ctypes as C_
        eng.lib.sperrhip_debug_counter.restype = C_.c_ulonglong
        eng.lib.sperrhip_debug_counter.argtypes = [C_.c_int]
        arena = int(eng.lib.sperrhip_debug_counter(3))
        workspace = {"arena_GB": round(arena / 1e9, 2), "MB_per_chunk_in_flight": round(arena / nchunks / 1e6, 1),
                     # what the chunk farm's workers hold after the host-path runs (all devices of the launch:
                     # two slots per worker, two workers per device)
                     "farm_pinned_staging_GB": round(int(eng.lib.sperrhip_debug_counter(4)) / 1e9, 3),
                     "farm_device_buffers_GB": round(int(eng.lib.sperrhip_debug_counter(5)) / 1e9, 3),
                     "what": "largest workspace arena of an engine after the timed steps (all chunks of the volume "
                             "in flight: the larger of the compression and the decompression layout)"}
    except Exception as ex:   # noqa: BLE001
        workspace = {"error": f"{type(ex).__name__}: {ex}"}

    # ---- the farm's host-thread plan against the CPUs this process may use (affinity mask, cgroup quota), for the
    #      devices of this launch and for a full node of 8
    farm_threads = None
    try:
        import ctypes as C_
        from sperr_amd.api import host_cpus
        eng.lib.sperrhip_farm_threads.argtypes = [C_.c_size_t, C_.c_size_t] + [C_.POINTER(C_.c_size_t)] * 5
        farm_threads = {"host": host_cpus(eng.lib)}
        for nd in sorted({max(1, world), 8}):
            v = [C_.c_size_t(0) for _ in range(5)]
            if eng.lib.sperrhip_farm_threads(0, nd, *[C_.byref(x) for x in v]) == 0:
                farm_threads[f"{nd}_devices"] = dict(zip(("workers_per_device", "dec_workers_per_device", "helpers_per_worker",
                                                          "threads_total", "cpus_usable"), (x.value for x in v)))
    except Exception as ex:   # noqa: BLE001
        farm_threads = {"error": f"{type(ex).__name__}: {ex}"}

    # ---- a small batch (rank 0): 8 chunks = 512^3, what one GPU gets when config 3's 64 chunks are
    #      dealt to 8 GPUs (strong scaling); device-resident like `value`
    small = None
    if not args.no_small_batch and S >= 2 * C:
        try:
            sv = vol[: 2 * C, : 2 * C, : 2 * C].contiguous()
            ss_ = eng.compress(sv, chunks, args.bpp).clone()
            so_ = eng.decompress(ss_, output_float=True)
            torch.cuda.synchronize()
            best_c, best_d = 1e9, 1e9
            for _ in range(5):
                torch.cuda.synchronize()
                a = time.perf_counter()
                eng.compress(sv, chunks, args.bpp)
                torch.cuda.synchronize()
                b = time.perf_counter()
                so_ = eng.decompress(ss_, output_float=True)
                torch.cuda.synchronize()
                c = time.perf_counter()
                best_c, best_d = min(best_c, b - a), min(best_d, c - b)
            small = {
                "what": f"{2 * C}^3 fp32 corner of the bench volume = 8 chunks of {C}^3 at BPP {args.bpp} (one GPU's share of "
                        "64 chunks on 8 GPUs), volume and container resident in HBM; best of 5",
                "compress_GBps": round(sv.numel() * 4 / best_c / 1e9, 3),
                "decompress_GBps": round(sv.numel() * 4 / best_d / 1e9, 3),
                "compress_ms": round(best_c * 1e3, 2), "decompress_ms": round(best_d * 1e3, 2),
                "decoded_identical_to_big_batch": bool(torch.equal(so_, out[: 2 * C, : 2 * C, : 2 * C])),
            }
            del sv, ss_, so_
        except Exception as ex:
            small = {"error": f"{type(ex).__name__}: {ex}"}

    # ---- a volume the chunk size does not divide (rank 0; reported beside the metric, never as it):
    #      chunk_volume (src/sperr_helper.cpp:542-592) leaves border chunks whose extents are not
    #      powers of two; their lists mix set shapes and decode through k_lis_mx
    ragged = None
    if not args.no_ragged and args.ragged_size > C:
        try:
            R = args.ragged_size
            del out
            rv = turbulence_torch((R, R, R), dev, seed=7)
            rs = eng.compress(rv, chunks, args.bpp).clone()
            ro = eng.decompress(rs, output_float=True)
            torch.cuda.synchronize()
            best_c, best_d = 1e9, 1e9
            for _ in range(2):
                torch.cuda.synchronize()
                a = time.perf_counter()
                eng.compress(rv, chunks, args.bpp)
                torch.cuda.synchronize()
                b = time.perf_counter()
                ro = eng.decompress(rs, output_float=True)
                torch.cuda.synchronize()
                c = time.perf_counter()
                best_c, best_d = min(best_c, b - a), min(best_d, c - b)
            per = [R // C + (1 if R % C > C // 2 else 0)] * 3
            ragged = {
                "what": f"{R}^3 fp32 in {C}^3 chunks at BPP {args.bpp}: {per[0] ** 3} chunks, all but {(R // C) ** 3} of them "
                        "with extents that are not powers of two; volume and container resident in HBM; best of 2",
                "compress_GBps": round(rv.numel() * 4 / best_c / 1e9, 3),
                "decompress_GBps": round(rv.numel() * 4 / best_d / 1e9, 3),
                "compress_ms": round(best_c * 1e3, 2), "decompress_ms": round(best_d * 1e3, 2),
                "max_abs_err": float((ro.double() - rv.double()).abs().max().item()),
            }
            del rv, rs, ro
        except Exception as ex:
            ragged = {"error": f"{type(ex).__name__}: {ex}"}

    # ---- the other modes of the path (rank 0; reported beside the metric, never as it): the bench
    #      volume in point-wise error mode (mode 3: the SPECK1D coder of the outliers on top of the
    #      chunk pipeline), and BASELINE configs[3]'s shape, one 999 x 999 slice at PSNR 90 dB through
    #      the 2D coder
    other = None
    if not args.no_other_modes:
        try:
            other = {}
            rngv = float(vol.max() - vol.min())
            tol = 1e-3 * rngv
            obuf = torch.empty(eng.max_compressed_size(vol.shape, chunks, tol, 3), dtype=torch.uint8, device=dev)
            ps = eng.compress(vol, chunks, tol, mode=3, out=obuf)
            po = eng.decompress(ps, output_float=True)
            torch.cuda.synchronize()
            best_c, best_d = 1e9, 1e9
            for _ in range(2):
                torch.cuda.synchronize()
                a = time.perf_counter()
                ps = eng.compress(vol, chunks, tol, mode=3, out=obuf)
                torch.cuda.synchronize()
                b = time.perf_counter()
                po = eng.decompress(ps, output_float=True)
                torch.cuda.synchronize()
                c = time.perf_counter()
                best_c, best_d = min(best_c, b - a), min(best_d, c - b)
            try:   # (everything the engine holds after a PWE volume: the arena, the slots and the mode's own buffers)
                pwe_dev = int(eng.lib.sperrhip_debug_counter(6))
            except Exception:   # noqa: BLE001
                pwe_dev = 0
            other["pwe_volume"] = {
                "engine_device_GB": round(pwe_dev / 1e9, 2),
                "engine_MB_per_chunk_in_flight": round(pwe_dev / nchunks / 1e6, 1),
                "what": f"the bench volume in point-wise error mode, tolerance 1e-3 of the range ({tol:.4g}); best of 2",
                "compress_GBps": round(vol.numel() * 4 / best_c / 1e9, 3),
                "decompress_GBps": round(vol.numel() * 4 / best_d / 1e9, 3),
                "bpp": round(ps.numel() * 8 / vol.numel(), 3),
                "max_abs_err": float((po.double() - vol.double()).abs().max().item()),
                # (the decoder narrows to fp32: half an fp32 ulp of the largest value comes on top)
                "within_tolerance": bool(float((po.double() - vol.double()).abs().max().item())
                                         <= tol + 6e-8 * float(vol.abs().max())),
            }
            del obuf, ps, po
            img = turbulence_torch((1, 999, 999), dev, seed=3)[0].contiguous()
            ss = eng.compress_2d(img, 90.0, mode=2).clone()
            so = eng.decompress_2d(ss, (999, 999), True)
            torch.cuda.synchronize()
            best_c, best_d = 1e9, 1e9
            for _ in range(3):
                torch.cuda.synchronize()
                a = time.perf_counter()
                eng.compress_2d(img, 90.0, mode=2)
                torch.cuda.synchronize()
                b = time.perf_counter()
                so = eng.decompress_2d(ss, (999, 999), True)
                torch.cuda.synchronize()
                c = time.perf_counter()
                best_c, best_d = min(best_c, b - a), min(best_d, c - b)
            mse = float(((so.double() - img.double()) ** 2).mean().item())
            rng2 = float(img.max() - img.min())
            other["slice_2d"] = {
                "what": "one 999 x 999 fp32 slice, PSNR 90 dB (sperr_comp_2d / sperr_decomp_2d on device buffers); best of 3",
                "compress_ms": round(best_c * 1e3, 3), "decompress_ms": round(best_d * 1e3, 3),
                "bpp": round(ss.numel() * 8 / img.numel(), 3),
                "psnr_dB": round(10 * __import__("math").log10(rng2 * rng2 / mse), 2),
            }
        except Exception as ex:
            other = {"error": f"{type(ex).__name__}: {ex}"}

    # ---- roofline of the dominant kernel (HIP events recorded by the engine, timed region) ----
    kern = sorted(prof_all.items(), key=lambda kv: -kv[1][2])   # (one untimed step, all kernels)
    # (the timed steps) top_ms: time during which the kernel was running -- decoding enqueues
    # sub-batches on several streams whose launches of one kernel overlap; top_sum: plain sum of
    # the launch durations, whose mean is what a kernel trace reports as the average duration
    # `achieved` uses the SUM: algorithmic bytes of a step / (launches per step x avg_launch_ms), so that
    # frac follows from the line's own avg_launch_ms and from a rocprofv3 kernel trace; the busy
    # (union) time is reported beside it
    top_ms, top_launches, top_sum = prof[top_name]
    busy_per_step_ms = top_ms / args.steps
    per_step_ms = top_sum / args.steps
    values = vol.numel()
    achieved = ALGO_BYTES_PER_VALUE * values / (per_step_ms / 1e3) / 1e9
    # HBM traffic of that kernel per step: PMC counters cannot be collected from inside this
    # process; profiles/*_pmc_traffic.json holds them for this very workload (separate rocprofv3
    # --pmc FETCH_SIZE / WRITE_SIZE passes, corrected as MI355X_MICROARCH.md prescribes)
    traffic = traffic_from = None
    if S == 1024 and C == 256 and args.bpp == 2.0:
        import glob
        from sperr_amd.srchash import bench_path_hash
        now = bench_path_hash()
        # only a record taken from THESE sources counts (the source hash decides; the order of the search -- by file
        # name, latest round's tag first -- only matters for speed: mtimes mean nothing in a fresh checkout); none:
        # null, and the line says why
        for pf in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), reverse=True):
            with open(pf) as f:
                doc = json.load(f)
            rec = doc["kernels"].get(top_name)
            if rec and doc.get("source_sha16") == now:
                traffic = rec["hbm_bytes_per_step_corrected"]
                traffic_from = {"file": os.path.basename(pf), "commit": doc.get("commit"), "source_sha16": now,
                                "step_total_GB": doc.get("total_hbm_GB_per_step")}
                break
        if traffic is None:
            traffic_from = {"file": None, "source_sha16": now,
                            "why": "no profiles/*_pmc_traffic.json was collected from the sources of this build"}
    roofline = {
        "bound": "hbm", "kernel": top_name, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_from": traffic_from,
        "algorithmic_bytes_per_step": ALGO_BYTES_PER_VALUE * values,
        "launches_per_step": top_launches // args.steps,
        "avg_launch_ms": round(top_sum / max(1, top_launches), 4),
        "kernel_ms_per_step": round(per_step_ms, 3),
        "kernel_busy_ms_per_step": round(busy_per_step_ms, 3),
        "top5_ms_per_step": {k: round(v[2], 3) for k, v in kern[:5]},
        # the whole step against the same peak: compress reads 4 B and writes BPP/8 B per value, decompress
        # the mirror (2 x 4.25 B per value per step) over the step's wall time
        "step_algorithmic_bytes": 2 * ALGO_BYTES_PER_VALUE * values,
        "step_achieved": round(2 * ALGO_BYTES_PER_VALUE * values / dt_max / 1e9, 3),
        "step_frac": round(2 * ALGO_BYTES_PER_VALUE * values / dt_max / 1e9 / HBM_PEAK_GBS, 6),
        "compress_frac": round(ALGO_BYTES_PER_VALUE * values / tc / 1e9 / HBM_PEAK_GBS, 6),
        "decompress_frac": round(ALGO_BYTES_PER_VALUE * values / td / 1e9 / HBM_PEAK_GBS, 6),
        # kernel launches of one step (the untimed step in which every kernel was bracketed)
        "launches_per_step_all_kernels": int(sum(v[1] for v in prof_all.values())),
    }

    if args.profile_out:
        with open(args.profile_out, "w") as f:
            # busy: time with at least one launch of the kernel running (the decoder's sub-batches
            # overlap on separate streams); avg: mean duration of one launch, as a trace reports it
            f.write("kernel,busy_ms_per_step,launches_per_step,avg_launch_ms,sum_ms_per_step\n")
            for k, (ms, cnt, sm) in kern:
                f.write(f"{k},{ms:.4f},{cnt},{sm / max(1, cnt):.5f},{sm:.4f}\n")

    # ---- CPU baseline: the same workload on this host's cores --------------------------------
    # The reference's chunk loop is `omp parallel for` over chunks (SPERR3D_OMP_C.cpp:94), so a
    # volume of 64 chunks keeps at most 64 threads busy: the sample is the bench volume itself
    # (or its --cpu-sample^3 corner) and `effective_threads` says how many threads had a chunk.
    # A one-thread figure on one chunk goes with it.
    cpu = None
    if not args.no_cpu_baseline:
        from oracle import pyoracle
        n = min(args.cpu_sample, S)
        sample = vol[:n, :n, :n].contiguous().cpu().numpy()
        # threads: what this process MAY use -- min(affinity mask, cgroup CFS quota), not os.cpu_count() (the
        # pool's GPU box shows 256 CPUs and grants the pod 16: 64 OpenMP threads on 16 CPUs of quota is a
        # throttled run, round 4's mistake) -- and no more than there are chunks, since the reference's chunk
        # loop gives a thread one chunk at a time (SPERR3D_OMP_C.cpp:94)
        from sperr_amd.api import host_cpus, host_throttle
        hc = host_cpus(eng.lib)
        nch = max(1, (n // C)) ** 3
        cores = max(1, min(hc["usable"], nch))
        if pyoracle.have_ref():
            impl, kind = pyoracle.Ref(), "reference"
        else:
            impl, kind = pyoracle.Oracle(), "port"
        thr_a = host_throttle(eng.lib)
        a = time.perf_counter()
        cs = impl.comp_3d(sample, chunks, 1, args.bpp, nthreads=cores)
        b = time.perf_counter()
        impl.decomp_3d(cs, True, nthreads=cores)
        c = time.perf_counter()
        thr_b = host_throttle(eng.lib)
        # parity of the timed data: the HIP container of the same sample must be byte-identical
        if n == S:
            hs = bytes(stream.cpu().numpy())
        else:
            hs = bytes(eng.compress(torch.from_numpy(sample).to(dev), chunks, args.bpp).cpu().numpy())
        one = sample[:C, :C, :C].copy() if n >= C else sample
        a1 = time.perf_counter()
        c1 = impl.comp_3d(one, chunks, 1, args.bpp, nthreads=1)
        b1 = time.perf_counter()
        impl.decomp_3d(c1, True, nthreads=1)
        d1 = time.perf_counter()
        cpu = {
            "value": round(sample.nbytes / (c - a) / 1e9, 4), "unit": "GB/s", "cores": cores,
            "nthreads_used": cores, "cores_visible": hc["cores_visible"], "affinity_cpus": hc["affinity"],
            "cpu_quota": hc["cpu_quota"],
            "cfs_throttle": (None if thr_a is None or thr_b is None else
                             {"nr_throttled": thr_b[0] - thr_a[0], "throttled_ms": round((thr_b[1] - thr_a[1]) / 1e3, 2)}),
            "kind": kind,
            "sample": f"{n}^3 fp32 of the bench volume ({nch} chunks of {C}^3), bpp {args.bpp}, {cores} OpenMP "
                      f"threads = min(CPUs this process may use: {hc['usable']} of {hc['cores_visible']} visible, "
                      f"chunks: {nch}): compress {b - a:.2f} s + decompress {c - b:.2f} s",
            "compress_GBps": round(sample.nbytes / (b - a) / 1e9, 4),
            "decompress_GBps": round(sample.nbytes / (c - b) / 1e9, 4),
            "one_thread": {"sample": f"one {C}^3 chunk", "compress_GBps": round(one.nbytes / (b1 - a1) / 1e9, 4),
                           "decompress_GBps": round(one.nbytes / (d1 - b1) / 1e9, 4),
                           "value": round(one.nbytes / (d1 - a1) / 1e9, 4)},
            "hip_stream_identical": hs == cs,
        }

    line = {
        "metric": "compress+decompress GB/s/GPU on 256^3 fp32 chunks @ BPP=2.0; bitstream-exact vs ref",
        "value": round(world * nbytes / dt_max / 1e9, 4),
        "unit": "GB/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt_max * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"synthetic {S}^3 fp32 turbulence per GPU, {C}^3 chunks, BPP={args.bpp}, "
                               "step = compress + decompress, volume and container resident in HBM",
                   "chunks_per_gpu": nchunks, "parallelism": f"chunks farmed over {world} GPU(s), no collective"},
        "compress_GBps_per_gpu": round(nbytes / tc / 1e9, 4),
        "decompress_GBps_per_gpu": round(nbytes / td / 1e9, 4),
        "container_bytes": int(stream.numel()), "container_len_exact": ok_len,
        "max_abs_err": err,
        "roofline": roofline,
        "cpu_baseline": cpu,
        "host_path": host_path,
        "host_path_pwe": host_path_pwe,
        # config 3 dealt to the N devices of this launch (one volume, strong scaling), H2D/D2H inside
        "strong_compress_GBps": (host_path or {}).get("compress_GBps"),
        "strong_decompress_GBps": (host_path or {}).get("decompress_GBps"),
        "small_batch": small,
        # what 8 GPUs can reach on ONE 1024^3 volume (config 3 dealt 8 chunks per GPU) if every GPU runs at this
        # GPU's 8-chunk rate and nothing else limits: the expectation a measured 8-GPU strong-scaling run is judged by
        "strong_ceiling_from_small_batch": (None if not small or "error" in small else {
            "n_gpus": 8,
            "compress_GBps": round(8 * small["compress_GBps"], 1), "decompress_GBps": round(8 * small["decompress_GBps"], 1),
            "x_one_gpu_compress": round(8 * small["compress_GBps"] / (nbytes / tc / 1e9), 2),
            "x_one_gpu_decompress": round(8 * small["decompress_GBps"] / (nbytes / td / 1e9), 2),
            "what": "8 x small_batch rate, device-resident; the host path adds PCIe (about 57 GB/s each way per GPU) and "
                    "the host's copy threads, which the farm sizes by the CPU quota (farm_threads)"}),
        "farm_threads": farm_threads,
        "workspace": workspace,
        "ragged_volume": ragged,
        "other_modes": other,
    }
    # ONE line either way.  The default is the compact one (under 4 KB: the driver's record keeps the tail of
    # stdout, and round 5's 12 KB line lost its host_path* fields there): every number, none of the prose --
    # --verbose prints the line with its "what" / "sample" descriptions, --full-out writes that one to a file.
    if args.full_out:
        with open(args.full_out, "w") as f:
            json.dump(line, f, indent=1)
    print(json.dumps(line if args.verbose else compact_line(line), separators=(",", ":")))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

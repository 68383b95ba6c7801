"""GPU: the in-library chunk farm and the streamed host path (sperr_amd/csrc/farm.hip) against the
oracle -- containers byte-identical, decoded volumes bit-identical.  The reference's equivalents
are the OpenMP chunk loops of /root/reference/src/SPERR3D_OMP_C.cpp:94-130 and
src/SPERR3D_OMP_D.cpp:101-127 behind sperr_comp_3d / sperr_decomp_3d
(/root/reference/src/SPERR_C_API.cpp:135-258).  One GPU is enough: a device may appear several
times in the device list, which gives it several workers with engines of their own."""
import os

import numpy as np
import pytest

from sperr_amd.synth import turbulence

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from sperr_amd.api import SperrHip
    return SperrHip()


class _Env:
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        os.environ.update({k: str(v) for k, v in self.kv.items()})

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


CASES = [
    ((96, 80, 72), (32, 32, 32), 1, 2.0),       # 3 x 2 x 2 grid with merged remainders: 4 shapes
    ((64, 64, 64), (32, 32, 32), 1, 4.0),
    ((50, 64, 72), (20, 30, 40), 2, 70.0),      # PSNR mode: stream lengths unknown beforehand
    ((72, 48, 40), (24, 24, 20), 3, 1e-3),      # PWE mode: outlier streams
    ((33, 17, 9), (64, 64, 64), 1, 3.0),        # one chunk
]


@pytest.mark.parametrize("shape,chunks,mode,quality", CASES)
@pytest.mark.parametrize("devices", [[0], [0, 0]])
def test_farm_matches_oracle(eng, oracle, shape, chunks, mode, quality, devices):
    vol = turbulence(shape)
    want = oracle.comp_3d(vol, chunks, mode, quality)
    got = eng.comp_3d_farm(vol, chunks, mode, quality, devices=devices)
    assert got == want
    back = eng.decomp_3d_farm(got, True, devices=devices)
    ref = oracle.decomp_3d(want, True)
    assert np.array_equal(back.view(np.uint32), ref.view(np.uint32))
    back64 = eng.decomp_3d_farm(got, False, devices=devices)
    ref64 = oracle.decomp_3d(want, False)
    assert np.array_equal(back64.view(np.uint64), ref64.view(np.uint64))


def test_streamed_one_chunk_per_item_pwe(eng, oracle):
    """The staging budget capped to one chunk per work item, so the volume has to stream through
    the device in 18 pieces (BASELINE config 5 in miniature: PWE mode + outlier coder); fp64 input."""
    vol = turbulence((72, 96, 64), dtype=np.float64)
    chunks, tol = (32, 32, 24), 1e-4
    want = oracle.comp_3d(vol, chunks, 3, tol)
    with _Env(SPERR_HIP_FARM_ITEM=1, SPERR_HIP_FARM_WORKERS=3):
        got = eng.comp_3d_farm(vol, chunks, 3, tol, devices=[0])
        assert got == want
        back = eng.decomp_3d_farm(got, False, devices=[0])
    assert np.array_equal(back.view(np.uint64), oracle.decomp_3d(want, False).view(np.uint64))
    assert np.abs(back - vol).max() <= tol


def test_reference_api_runs_on_the_farm(eng, oracle):
    """sperr_comp_3d / sperr_decomp_3d (the symbols H5Z-SPERR and the CLI bind) with helper threads."""
    vol = turbulence((80, 70, 60))
    want = oracle.comp_3d(vol, (32, 32, 32), 1, 2.5)
    assert eng.comp_3d(vol, (32, 32, 32), 1, 2.5, nthreads=6) == want
    back = eng.decomp_3d(want, True, nthreads=6)
    assert np.array_equal(back.view(np.uint32), oracle.decomp_3d(want, True).view(np.uint32))


@pytest.mark.parametrize("copy", ["3d", "stage"])
def test_pinned_volume_dma(eng, oracle, copy):
    """A caller's pinned volume is read / written by DMA (no staging copy) -- or staged when asked."""
    import torch
    vol = turbulence((64, 96, 80))
    pinned = torch.from_numpy(vol).pin_memory()
    want = oracle.comp_3d(vol, (32, 32, 32), 1, 2.0)
    with _Env(SPERR_HIP_PINNED_COPY=copy):
        assert eng.comp_3d_farm(pinned, (32, 32, 32), 1, 2.0, devices=[0, 0]) == want
        out = torch.empty(vol.shape, dtype=torch.float32).pin_memory()
        eng.decomp_3d_into(want, out, devices=[0, 0])
    assert np.array_equal(out.numpy().view(np.uint32), oracle.decomp_3d(want, True).view(np.uint32))


def test_truncated_container_through_the_farm(eng, oracle):
    vol = turbulence((64, 64, 48))
    full = oracle.comp_3d(vol, (32, 32, 24), 1, 4.0)
    part = eng.trunc_3d(full, 40)
    back = eng.decomp_3d_farm(part, True, devices=[0, 0])
    assert np.array_equal(back.view(np.uint32), oracle.decomp_3d(part, True).view(np.uint32))


def test_farm_rejects_bad_arguments(eng):
    import ctypes as C
    vol = turbulence((16, 16, 16))
    dst, n = C.c_void_p(None), C.c_size_t(0)
    f = eng.lib.sperrhip_comp_3d_farm
    assert f(vol.ctypes.data, 1, 16, 16, 16, 8, 8, 8, 1, -1.0, 0, None, 0, C.byref(dst), C.byref(n)) == 2
    assert f(vol.ctypes.data, 1, 16, 16, 16, 8, 8, 8, 7, 2.0, 0, None, 0, C.byref(dst), C.byref(n)) == 2
    bad = (C.c_int * 1)(99)
    assert f(vol.ctypes.data, 1, 16, 16, 16, 8, 8, 8, 1, 2.0, 0, bad, 1, C.byref(dst), C.byref(n)) == -1
    assert not dst.value
    with pytest.raises(Exception):
        eng.decomp_3d_farm(b"\x00" * 64, True)


def test_device_pointers_through_the_host_entry_points(eng, oracle):
    """A volume that already lives on a device (or is wanted there) is not an error for
    sperrhip_comp_3d_farm / sperrhip_decomp_3d_into: the device-resident path runs where the data is."""
    import ctypes as C
    import torch
    vol = turbulence((48, 64, 80))
    want = oracle.comp_3d(vol, (32, 32, 32), 1, 2.0)
    dv = torch.from_numpy(vol).cuda()
    dst, n = C.c_void_p(None), C.c_size_t(0)
    rc = eng.lib.sperrhip_comp_3d_farm(dv.data_ptr(), 1, 80, 64, 48, 32, 32, 32, 1, 2.0, 0, None, 0,
                                       C.byref(dst), C.byref(n))
    assert rc == 0
    got = C.string_at(dst.value, n.value)
    eng._libc.free(dst)
    assert got == want
    out = torch.empty((48, 64, 80), dtype=torch.float32, device="cuda")
    buf = np.frombuffer(want, dtype=np.uint8)
    d = [C.c_size_t(0) for _ in range(3)]
    rc = eng.lib.sperrhip_decomp_3d_into(buf.ctypes.data, buf.size, 1, 0, None, 0, out.data_ptr(), out.numel() * 4,
                                         C.byref(d[0]), C.byref(d[1]), C.byref(d[2]))
    assert rc == 0 and (d[0].value, d[1].value, d[2].value) == (80, 64, 48)
    assert np.array_equal(out.cpu().numpy().view(np.uint32), oracle.decomp_3d(want, True).view(np.uint32))


def test_config5_shape_pinned_pwe_256_cubes_streamed(eng, oracle):
    """BASELINE config 5 at its own shape on one GPU: a PINNED fp32 volume of 16 chunks of 256^3
    (512 x 512 x 1024), point-wise error mode with the outlier coder, streamed through the farm in
    work items of 2 chunks by 3 workers, the engines' workspace capped (SPERR_HIP_ARENA_MAX_MB) so
    that neither the volume's fp64 chunk buffers nor a whole item's workspace could be resident at
    once: container byte-identical to the oracle's (its OpenMP chunk loop on all host threads, the
    reference's src/SPERR3D_OMP_C.cpp:94-130 + src/SPECK_FLT.cpp:461-486), decoded floats
    bit-identical, tolerance met -- and the same once from pageable memory."""
    import torch
    from sperr_amd.synth import turbulence_torch
    shape, chunks = (1024, 512, 512), (256, 256, 256)
    dvol = turbulence_torch(shape, torch.device("cuda", 0))
    span = float(dvol.max() - dvol.min())
    tol = 1e-3 * span
    pinned = torch.empty(shape, dtype=torch.float32).pin_memory()
    pinned.copy_(dvol)
    del dvol
    torch.cuda.empty_cache()
    vol = pinned.numpy()
    nthreads = os.cpu_count() or 8
    want = oracle.comp_3d(vol, chunks, 3, tol, nthreads=nthreads)
    ref = oracle.decomp_3d(want, True, nthreads=nthreads)
    # (narrowing the decoded doubles to fp32 adds up to half an fp32 ulp of the value)
    assert float(np.abs(ref.astype(np.float64) - vol).max()) <= tol + 6e-8 * float(np.abs(vol).max())
    # every chunk carries an outlier stream at this tolerance? at least most of them must, or the
    # outlier coder is not what is being tested
    hdr = 20 + 4 * 16
    lens = np.frombuffer(want[20:hdr], dtype=np.uint32).astype(np.int64)
    at, with_outliers = hdr, 0
    for l in lens:
        tb = int(np.frombuffer(want[at + 18:at + 26], dtype=np.uint64)[0])
        with_outliers += int(26 + (tb + 7) // 8 + 9 < l)
        at += int(l)
    assert with_outliers >= 8, f"only {with_outliers} of 16 chunks have outliers at tol {tol}"

    with _Env(SPERR_HIP_FARM_ITEM=2, SPERR_HIP_FARM_WORKERS=3, SPERR_HIP_FARM_DEC_WORKERS=3,
              SPERR_HIP_ARENA_MAX_MB=1024):
        eng.release()                       # engines sized by earlier tests start from nothing
        got = eng.comp_3d_farm(pinned, chunks, 3, tol, devices=[0])
        assert len(got) == len(want) and got == want
        out = torch.empty(shape, dtype=torch.float32).pin_memory()
        eng.decomp_3d_into(got, out, devices=[0])
        assert np.array_equal(out.numpy().view(np.uint32), ref.view(np.uint32))
        # pageable: rows staged by the helper threads
        pageable = vol.copy()
        assert eng.comp_3d_farm(pageable, chunks, 3, tol, devices=[0]) == want
        pout = np.empty(shape, dtype=np.float32)
        eng.decomp_3d_into(got, pout, devices=[0])
        assert np.array_equal(pout.view(np.uint32), ref.view(np.uint32))
    eng.release()


def _hip_fns(eng):
    import torch

    def comp(v, c, m, q):
        return bytes(eng.compress(torch.from_numpy(np.ascontiguousarray(v)).cuda(), c, q, mode=m).cpu().numpy())

    def decomp(s, as_float):
        return eng.decompress(torch.from_numpy(np.frombuffer(s, dtype=np.uint8).copy()).cuda(), as_float).cpu().numpy()

    return comp, decomp


@pytest.mark.parametrize("shape,chunks,mode,quality",
                         [((96, 80, 72), (32, 32, 32), 1, 2.0), ((72, 48, 40), (24, 24, 20), 3, 1e-3)])
def test_process_per_gpu_farm_with_the_hip_compressor(eng, oracle, shape, chunks, mode, quality):
    """sperr_amd/farm.py (the one-process-per-GPU dealing of bench.py --gpus N) with the HIP engine as
    the per-rank compressor, world size 1: the stitched container and the scattered volume are the
    oracle's.  (tests/test_farm_gloo.py runs the same module on two CPU ranks with the oracle.)"""
    from sperr_amd import farm
    comp, decomp = _hip_fns(eng)
    vol = turbulence(shape)
    want = oracle.comp_3d(vol, chunks, mode, quality)
    assert farm.farm_compress(vol, chunks, quality, comp, mode=mode) == want
    back = farm.farm_decompress(want, decomp, True)
    assert np.array_equal(back.view(np.uint32), oracle.decomp_3d(want, True).view(np.uint32))


def _rank_main(rank, world, port, shape, chunks, mode, quality, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sperr_amd import farm
    from sperr_amd.api import SperrHip
    comp, decomp = _hip_fns(SperrHip())
    vol = turbulence(shape)
    out = farm.farm_compress(vol, chunks, quality, comp, mode=mode)
    # every rank needs the container to decode its share: rank 0 hands it out
    box = [out]
    dist.broadcast_object_list(box, src=0)
    back = farm.farm_decompress(box[0], decomp, True)
    if rank == 0:
        q.put((out, back.tobytes()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_share_the_device_with_the_hip_compressor(oracle):
    """Two ranks (gloo for the host-side gather, both on cuda:0): each compresses and decodes its
    half of the chunks on the GPU; rank 0 stitches.  No data-path collective."""
    import socket
    import torch.multiprocessing as mp
    shape, chunks, mode, quality = (64, 64, 96), (32, 32, 32), 1, 2.0
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, shape, chunks, mode, quality, q)) for r in range(2)]
    for p in procs:
        p.start()
    merged, back = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    vol = turbulence(shape)
    want = oracle.comp_3d(vol, chunks, mode, quality)
    assert merged == want
    assert back == oracle.decomp_3d(want, True).tobytes()


def test_distinct_devices_when_the_box_has_them(eng, oracle):
    """The farm over TWO DIFFERENT devices (round 4): runs only where `torch.cuda.device_count() >= 2` -- the pool's
    one-GPU boxes skip it, the first multi-GPU box exercises it.  No peer access, no collective: each
    device's workers stream their items through their own staging buffers; the container and the decoded
    volume must be the oracle's whichever device took which item.  All three modes, pageable and pinned."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one visible device: the farm's multi-device path needs two")
    devices = list(range(min(torch.cuda.device_count(), 4)))
    for shape, chunks, mode, quality in CASES[:4]:
        vol = turbulence(shape)
        want = oracle.comp_3d(vol, chunks, mode, quality)
        for pinned in (False, True):
            src = torch.from_numpy(vol).pin_memory() if pinned else vol
            with _Env(SPERR_HIP_FARM_ITEM=1):     # one chunk per item: every device gets work
                got = eng.comp_3d_farm(src, chunks, mode, quality, devices=devices)
                assert got == want, (shape, mode, pinned)
                back = eng.decomp_3d_farm(got, True, devices=devices)
            assert np.array_equal(back.view(np.uint32), oracle.decomp_3d(want, True).view(np.uint32))
    # every listed device has a place the farm could look up (node may be -1 in a VM)
    import ctypes as C
    for d in devices:
        bdf = C.create_string_buffer(64)
        node, n = C.c_int(-9), C.c_size_t(0)
        assert eng.lib.sperrhip_farm_device_place(d, bdf, 64, C.byref(node), C.byref(n)) == 0
        assert bdf.value and node.value >= -1


def test_farm_reports_the_place_of_device_0(eng):
    """sperrhip_farm_device_place on the real sysfs of the GPU box: a PCI address in the usual spelling and,
    where the platform names a node, a non-empty CPU list (the workers of the tests above ran bound to it)."""
    import ctypes as C
    import re
    bdf = C.create_string_buffer(64)
    node, n = C.c_int(-9), C.c_size_t(0)
    assert eng.lib.sperrhip_farm_device_place(0, bdf, 64, C.byref(node), C.byref(n)) == 0
    assert re.fullmatch(r"[0-9a-fA-F]{4}:[0-9a-fA-F]{2}:[0-9a-fA-F]{2}\.[0-7]", bdf.value.decode())
    assert node.value >= -1
    if node.value >= 0:
        assert n.value > 0
    assert eng.lib.sperrhip_farm_device_place(10 ** 6, bdf, 64, C.byref(node), C.byref(n)) == -1

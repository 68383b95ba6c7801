"""CPU: the queue of the in-library chunk farm (sperr_amd/csrc/farm.hip) without any device --
how a volume's chunks are cut into work items and how idle workers share them
(sperrhip_farm_selftest).  Replaces the OpenMP chunk loop of
/root/reference/src/SPERR3D_OMP_C.cpp:94-130 (dynamic schedule over chunks)."""
import ctypes as C

import numpy as np
import pytest

from sperr_amd import api, farm


def _selftest(vol_xyz, chunk_xyz, esz, ndev, wpd, lockstep=1):
    lib = api.load_library()
    lib.sperrhip_farm_selftest.restype = C.c_int
    lib.sperrhip_farm_selftest.argtypes = [C.c_size_t] * 9 + [C.c_int] + [C.c_void_p] * 4
    nchunks = len(farm.chunk_grid(vol_xyz[::-1], chunk_xyz))
    per_worker = np.zeros(ndev * wpd, dtype=np.uint32)
    item_of = np.full(nchunks, 0xffffffff, dtype=np.uint32)
    worker_of = np.full(nchunks, 0xffffffff, dtype=np.uint32)
    nitems = C.c_size_t(0)
    rc = lib.sperrhip_farm_selftest(*vol_xyz, *chunk_xyz, esz, ndev, wpd, lockstep,
                                    per_worker.ctypes.data, item_of.ctypes.data, worker_of.ctypes.data,
                                    C.byref(nitems))
    assert rc == 0
    return per_worker, item_of, worker_of, nitems.value


def test_config3_64_chunks_over_8_devices():
    """BASELINE config 3: 1024^3 fp32 in 256^3 chunks on 8 devices -- every chunk is handed out
    exactly once and no device gets more than its 8."""
    per_worker, item_of, worker_of, nitems = _selftest((1024,) * 3, (256,) * 3, 4, 8, 1)
    assert per_worker.sum() == 64 and per_worker.max() <= 8 and per_worker.min() >= 8
    assert (item_of != 0xffffffff).all() and (worker_of < 8).all()
    # an item is a run of chunks in chunk_volume order
    for it in range(nitems):
        ids = np.nonzero(item_of == it)[0]
        assert len(ids) and (np.diff(ids) == 1).all()
    # several workers per device (worker w drives device w mod 8): still at most 8 per device... in
    # lockstep every worker takes one item per round
    per_worker, _, worker_of, _ = _selftest((1024,) * 3, (256,) * 3, 4, 8, 3)
    assert per_worker.sum() == 64
    per_dev = np.bincount(worker_of % 8, minlength=8)
    assert per_dev.max() <= 12 and per_dev.min() >= 4


@pytest.mark.parametrize("vol,chunk,ndev,wpd", [((700, 600, 1300), (128, 128, 128), 8, 3),
                                                ((40, 48, 56), (32, 32, 32), 1, 3),
                                                ((4096, 4096, 4096), (256, 256, 256), 8, 3),
                                                ((64, 64, 64), (16, 16, 16), 2, 2),
                                                ((33, 17, 9), (64, 64, 64), 8, 3)])
def test_every_chunk_once_items_of_one_shape(vol, chunk, ndev, wpd):
    grid = farm.chunk_grid(vol[::-1], chunk)
    for lockstep in (1, 0):
        per_worker, item_of, worker_of, nitems = _selftest(vol, chunk, 4, ndev, wpd, lockstep)
        assert per_worker.sum() == len(grid)
        assert (item_of < nitems).all() and (worker_of < ndev * wpd).all()
        for it in range(min(nitems, 200)):
            ids = np.nonzero(item_of == it)[0]
            shapes = {(grid[i][1], grid[i][3], grid[i][5]) for i in ids}
            assert len(shapes) == 1                    # equally shaped chunks only
            nbytes = len(ids) * np.prod(list(shapes)[0]) * 4
            assert nbytes <= (768 << 20) or len(ids) == 1   # staging buffers stay bounded


def test_config5_items_stream():
    """BASELINE config 5: 4096^3 fp32 = 4096 chunks of 256^3 (275 GB): no item is larger than the
    staging budget, so the volume streams through the devices."""
    per_worker, item_of, _, nitems = _selftest((4096,) * 3, (256,) * 3, 4, 8, 3)
    assert per_worker.sum() == 4096
    counts = np.bincount(item_of)
    assert counts.max() * 256 ** 3 * 4 <= 768 << 20
    assert nitems >= 8 * 3 * 2

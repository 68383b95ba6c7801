"""CPU, world_size 2, gloo: the one-process-per-GPU flavour of the chunk farm (sperr_amd/farm.py:
balanced runs of chunks per rank, equally shaped chunks stacked into small volumes, byte streams
gathered and stitched) with the oracle standing in for the per-rank GPU compressor.  The merged
container must be byte-identical to compressing the whole volume at once, and the farmed decode
bit-identical to decoding it at once.  The in-library farm (farm.hip) is covered by
tests/test_farm_queue.py (CPU) and tests/test_gpu_farm.py (GPU)."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from sperr_amd import farm
from sperr_amd.synth import turbulence


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, shape, chunks, mode, quality, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.pyoracle import Oracle
    orc = Oracle()
    vol = turbulence(shape)
    out = farm.farm_compress(vol, chunks, quality, lambda v, c, m, b: orc.comp_3d(v, c, m, b), mode=mode)
    whole = orc.comp_3d(vol, chunks, mode, quality)
    back = farm.farm_decompress(whole, lambda s, f: orc.decomp_3d(s, f))
    tmax = farm.reduce_max_seconds(1.0 + rank)
    if rank == 0:
        q.put((out, whole, back.tobytes(), orc.decomp_3d(whole, True).tobytes(), tmax))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("shape,chunks,mode,quality",
                         [((64, 32, 32), (16, 16, 16), 1, 2.0), ((50, 40, 24), (16, 16, 16), 1, 2.0),
                          ((32, 32, 32), (32, 32, 32), 1, 2.0), ((40, 40, 24), (16, 16, 16), 3, 1e-2)])
def test_farm_two_ranks_matches_single_pass(shape, chunks, mode, quality):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, shape, chunks, mode, quality, q)) for r in range(2)]
    for p in procs:
        p.start()
    merged, whole, back, back_whole, tmax = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert merged == whole
    assert back == back_whole
    assert tmax == 2.0


def test_dealing_and_segments():
    assert farm.chunk_segments(91, 64) == [(0, 91)]
    assert farm.chunk_segments(128, 64) == [(0, 64), (64, 64)]
    assert farm.chunk_segments(100, 30) == [(0, 30), (30, 30), (60, 40)]
    grid = farm.chunk_grid((1024, 1024, 1024), (256, 256, 256))
    assert len(grid) == 64 and grid[1] == (256, 256, 0, 256, 0, 256)
    deal = farm.deal_chunks(len(grid), 8)           # BASELINE config 3: 8 chunks on each of 8 GPUs
    assert [len(d) for d in deal] == [8] * 8
    assert sorted(i for d in deal for i in d) == list(range(64))
    assert [len(d) for d in farm.deal_chunks(4, 8)] == [0, 1, 0, 1, 0, 1, 0, 1]

"""CPU, world_size 2, gloo: the multi-GPU chunk farm's host logic (sharding z-slabs of chunks over
ranks, gathering the byte streams, stitching the container) with the oracle standing in for the
per-rank GPU compressor.  The merged container must be byte-identical to compressing the whole
volume at once."""
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


def _worker(rank, world, port, shape, chunks, bpp, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.pyoracle import Oracle
    orc = Oracle()
    vol = turbulence(shape)
    out = farm.farm_compress(vol, chunks, bpp, lambda v, c, b: orc.comp_3d(v, c, 1, b))
    tmax = farm.reduce_max_seconds(1.0 + rank)
    if rank == 0:
        q.put((out, orc.comp_3d(vol, chunks, 1, bpp), tmax))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("shape,chunks", [((64, 32, 32), (16, 16, 16)), ((50, 40, 24), (16, 16, 16)),
                                          ((32, 32, 32), (32, 32, 32))])
def test_farm_two_ranks_matches_single_pass(shape, chunks):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, shape, chunks, 2.0, q)) for r in range(2)]
    for p in procs:
        p.start()
    merged, whole, tmax = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert merged == whole
    assert tmax == 2.0


def test_shard_and_segments():
    assert farm.chunk_segments(91, 64) == [(0, 91)]
    assert farm.chunk_segments(128, 64) == [(0, 64), (64, 64)]
    assert farm.chunk_segments(100, 30) == [(0, 30), (30, 30), (60, 40)]
    assert farm.shard_z_slabs((1024, 1024, 1024), (256, 256, 256), 8)[:3] == [(0, 256), (256, 512), (512, 768)]
    assert farm.shard_z_slabs((1024, 8, 8), (256, 256, 256), 8)[4:] == [(1024, 1024)] * 4

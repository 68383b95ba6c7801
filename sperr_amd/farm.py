"""Chunk farming with one PROCESS per GPU (torch.distributed), for hosts that are organised that way.

The product farm lives in the library (sperr_amd/csrc/farm.hip: one process, worker threads per
device, a shared queue of work items; it is what sperr_comp_3d / sperr_decomp_3d run on).  This
module is the same dealing for ranks that each own one GPU: chunks are independent end to end
(separate mean, q, DWT, SPECK stream -- /root/reference/src/SPERR3D_OMP_C.cpp:94-130), so the data
path needs NO collective.  Every rank takes a balanced run of CHUNKS in chunk_volume order (x
fastest, then y, then z -- /root/reference/src/sperr_helper.cpp:579-589), stacks equally shaped
ones along z into a small volume, compresses that with its device compressor, and only the finished
byte streams travel (a host-side gather of variable-length blobs) to be stitched into one container
(layout: /root/reference/src/SPERR3D_OMP_C.cpp:163-234).  Decompression mirrors it.
"""
import struct

import numpy as np


def chunk_segments(vol_len, chunk_len):
    """Segment boundaries along one axis (src/sperr_helper.cpp:545-577): a remainder longer than
    half a chunk is its own segment, a shorter one is merged into the last segment."""
    n = vol_len // chunk_len
    if vol_len % chunk_len > chunk_len // 2:
        n += 1
    n = max(n, 1)
    tics = [i * chunk_len for i in range(n)] + [vol_len]
    return [(tics[i], tics[i + 1] - tics[i]) for i in range(n)]


def chunk_grid(vol_zyx, chunks_xyz):
    """[(x0, lx, y0, ly, z0, lz)] in chunk_volume order (x fastest)."""
    vz, vy, vx = vol_zyx
    cd = [min(max(c, 1), v) for c, v in zip(chunks_xyz, (vx, vy, vz))]
    sx, sy, sz = chunk_segments(vx, cd[0]), chunk_segments(vy, cd[1]), chunk_segments(vz, cd[2])
    return [(x0, lx, y0, ly, z0, lz) for z0, lz in sz for y0, ly in sy for x0, lx in sx]


def deal_chunks(nchunks, world):
    """Balanced contiguous runs of chunk indices, one list per rank (every rank gets work as long
    as there are at least `world` chunks: 64 chunks over 8 ranks = 8 each)."""
    return [list(range(nchunks * r // world, nchunks * (r + 1) // world)) for r in range(world)]


def split_container(stream):
    """-> (flags, vol_dims_xyz, chunk_dims_xyz, [chunk streams])"""
    flags = stream[1]
    vx, vy, vz = struct.unpack_from("<3I", stream, 2)
    pos = 14
    cd = (vx, vy, vz)
    if flags & 0x10:
        cd = struct.unpack_from("<3H", stream, 14)
        pos = 20
    n = 1
    for v, c in zip((vx, vy, vz), cd):
        n *= len(chunk_segments(v, c))
    lens = struct.unpack_from(f"<{n}I", stream, pos)
    pos += 4 * n
    parts = []
    for ln in lens:
        parts.append(stream[pos:pos + ln])
        pos += ln
    assert pos == len(stream)
    return flags, (vx, vy, vz), cd, parts


def build_container(parts, vol_xyz, chunk_xyz, is_float, portion=False):
    """Header (src/SPERR3D_OMP_C.cpp:163-234) + the chunk streams back to back."""
    multi = len(parts) > 1
    head = bytearray([0, 0x40 | (0x20 if is_float else 0) | (0x10 if multi else 0) | (0x80 if portion else 0)])
    head += struct.pack("<3I", *vol_xyz)
    if multi:
        head += struct.pack("<3H", *chunk_xyz)
    head += struct.pack(f"<{len(parts)}I", *[len(p) for p in parts])
    return bytes(head) + b"".join(parts)


def _groups(ids, grid):
    """the rank's chunks grouped by shape, order kept"""
    g = {}
    for i in ids:
        x0, lx, y0, ly, z0, lz = grid[i]
        g.setdefault((lx, ly, lz), []).append(i)
    return g


def _gather(obj, group=None):
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world == 1:
        return [obj]
    blobs = [None] * world if rank == 0 else None
    dist.gather_object(obj, blobs, dst=0, group=group)
    return blobs


def farm_compress(vol, chunks_xyz, quality, compress_fn, mode=1, group=None):
    """Every rank holds `vol` (numpy, shaped z,y,x) or at least its own chunks of it and compresses
    them with `compress_fn(subvol, chunks_xyz, mode, quality) -> bytes`; rank 0 returns the container
    of the whole volume, the other ranks None.  No collective on the data path; one gather of blobs."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    grid = chunk_grid(vol.shape, chunks_xyz)
    mine = {}
    for (lx, ly, lz), ids in _groups(deal_chunks(len(grid), world)[rank], grid).items():
        # equally shaped chunks stacked along z are a volume that is cut into exactly these chunks
        stack = np.concatenate([vol[grid[i][4]:grid[i][4] + lz, grid[i][2]:grid[i][2] + ly,
                                    grid[i][0]:grid[i][0] + lx] for i in ids], axis=0)
        parts = split_container(compress_fn(np.ascontiguousarray(stack), (lx, ly, lz), mode, quality))[3]
        assert len(parts) == len(ids)
        mine.update(zip(ids, parts))
    blobs = _gather(mine, group)
    if rank != 0:
        return None
    parts = {}
    for b in blobs:
        parts.update(b)
    vz, vy, vx = vol.shape
    cd = tuple(min(max(c, 1), v) for c, v in zip(chunks_xyz, (vx, vy, vz)))
    return build_container([parts[i] for i in range(len(grid))], (vx, vy, vz), cd, vol.dtype == np.float32)


def farm_decompress(stream, decompress_fn, output_float=True, group=None):
    """`decompress_fn(container_bytes, output_float) -> numpy (z,y,x)`; rank 0 returns the volume."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    flags, (vx, vy, vz), cd, parts = split_container(stream)
    grid = chunk_grid((vz, vy, vx), cd)
    mine = {}
    for (lx, ly, lz), ids in _groups(deal_chunks(len(grid), world)[rank], grid).items():
        mini = build_container([parts[i] for i in ids], (lx, ly, lz * len(ids)), (lx, ly, lz),
                               bool(flags & 0x20), bool(flags & 0x80))
        out = decompress_fn(mini, output_float)
        for k, i in enumerate(ids):
            mine[i] = np.ascontiguousarray(out[k * lz:(k + 1) * lz])
    blobs = _gather(mine, group)
    if rank != 0:
        return None
    vol = np.empty((vz, vy, vx), dtype=np.float32 if output_float else np.float64)
    for b in blobs:
        for i, a in b.items():
            x0, lx, y0, ly, z0, lz = grid[i]
            vol[z0:z0 + lz, y0:y0 + ly, x0:x0 + lx] = a
    return vol


def reduce_max_seconds(seconds, device=None, group=None):
    """max over ranks of a wall-time measurement (bench.py contract)."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([seconds], dtype=torch.float64, device=device or "cpu")
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())

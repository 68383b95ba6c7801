"""Chunk farming over the GPUs of one node (one process per GPU, torch.distributed).

Chunks are independent end to end (separate mean, q, DWT, SPECK stream --
/root/reference/src/SPERR3D_OMP_C.cpp:94-130), so the data path needs NO collective: every rank
compresses whole z-slabs of chunks of the volume on its own GPU, and only the finished byte
streams travel (a host-side gather of variable-length blobs) to be stitched into one container
in chunk_volume order (x fastest, then y, then z -- /root/reference/src/sperr_helper.cpp:579-589;
container layout /root/reference/src/SPERR3D_OMP_C.cpp:163-234).
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


def shard_z_slabs(vol_zyx, chunks_xyz, world):
    """Deal the z-segments of the chunk grid to `world` ranks in contiguous runs.  Returns a list
    of (z0, z1) voxel ranges, one per rank (z0 == z1 for a rank without work)."""
    segs = chunk_segments(vol_zyx[0], min(max(chunks_xyz[2], 1), vol_zyx[0]))
    n = len(segs)
    out, start = [], 0
    for r in range(world):
        cnt = n // world + (1 if r < n % world else 0)
        if cnt == 0:
            out.append((vol_zyx[0], vol_zyx[0]))
            continue
        z0 = segs[start][0]
        z1 = segs[start + cnt - 1][0] + segs[start + cnt - 1][1]
        out.append((z0, z1))
        start += cnt
    return out


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


def merge_containers(slab_streams, vol_zyx, chunks_xyz, is_float):
    """Stitch the per-slab containers (in rank order) into the container of the whole volume."""
    cd = tuple(min(max(c, 1), v) for c, v in zip(chunks_xyz, vol_zyx[::-1]))
    parts = []
    for s in slab_streams:
        if s:
            parts.extend(split_container(s)[3])
    multi = len(parts) > 1
    head = bytearray([0, 0x40 | (0x20 if is_float else 0) | (0x10 if multi else 0)])
    head += struct.pack("<3I", vol_zyx[2], vol_zyx[1], vol_zyx[0])
    if multi:
        head += struct.pack("<3H", *cd)
    head += struct.pack(f"<{len(parts)}I", *[len(p) for p in parts])
    return bytes(head) + b"".join(parts)


def farm_compress(vol, chunks_xyz, bpp, compress_fn, group=None):
    """Every rank holds (at least) its own z-slab of `vol` (numpy, shaped z,y,x) and compresses it
    with `compress_fn(subvol, chunks_xyz, bpp) -> bytes`; rank 0 returns the merged container,
    the other ranks return None.  No collective on the data path; one gather of byte blobs."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    slabs = shard_z_slabs(vol.shape, chunks_xyz, world)
    z0, z1 = slabs[rank]
    mine = b""
    if z1 > z0:
        # the slab is cut with the volume's own chunk dims so that its chunk grid is the
        # corresponding part of the volume's grid
        cd = tuple(min(max(c, 1), v) for c, v in zip(chunks_xyz, vol.shape[::-1]))
        mine = compress_fn(np.ascontiguousarray(vol[z0:z1]), cd, bpp)
    if world == 1:
        blobs = [mine]
    else:
        blobs = [None] * world if rank == 0 else None
        dist.gather_object(mine, blobs, dst=0, group=group)
    if rank != 0:
        return None
    return merge_containers(blobs, vol.shape, chunks_xyz, vol.dtype == np.float32)


def reduce_max_seconds(seconds, device=None, group=None):
    """max over ranks of a wall-time measurement (bench.py contract)."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([seconds], dtype=torch.float64, device=device or "cpu")
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())

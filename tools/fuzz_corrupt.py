"""Damaged containers: the decoder has to reject them or decode them to something, and stay usable.

    python tools/fuzz_corrupt.py [trials per case] [seed]

Every trial flips a few bytes of a valid container (3D in the three modes, a 2D slice stream) --
anywhere, or inside the headers only -- or cuts it short, and decodes it.  A return code or any
output is fine; a fault or a hang is not.  After every case the undamaged container is decoded again
and has to give the very same values as before.
"""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np
import torch

from sperr_amd.api import SperrHip, SperrHipError


def field(shape, seed):
    rng = np.random.default_rng(seed)
    z, y, x = np.meshgrid(*[np.linspace(0, 3, n) for n in shape], indexing="ij")
    return (np.sin(2 * x + y) * np.cos(z - x) + 0.05 * rng.standard_normal(shape)).astype(np.float32)


def chunk_headers(c):
    """Offsets of the header bytes inside a multi-chunk 3D container: 17 + 9 bytes in front of every
    chunk, 9 more in front of its outlier stream if it has one."""
    vol = np.frombuffer(c[2:14].tobytes(), dtype=np.uint32).astype(np.int64)
    chk = np.frombuffer(c[14:20].tobytes(), dtype=np.uint16).astype(np.int64)
    nseg = [max(1, int(v // k) + (1 if v % k > k // 2 else 0)) for v, k in zip(vol, chk)]
    n = nseg[0] * nseg[1] * nseg[2]
    lens = np.frombuffer(c[20:20 + 4 * n].tobytes(), dtype=np.uint32).astype(np.int64)
    at, idx = 20 + 4 * n, []
    for l in lens:
        idx += list(range(at, at + min(l, 26)))
        if l > 26 and not (c[at] & 1):
            bits = int(np.frombuffer(c[at + 18:at + 26].tobytes(), dtype=np.uint64)[0])
            first = 26 + (bits + 7) // 8
            if first + 9 <= l:
                idx += list(range(at + first, at + first + 9))
        at += l
    return np.array(idx)


def damage(rng, good, kind, heads=None):
    bad = good.copy()
    n = len(bad)
    if kind == 3 and heads is None:
        kind = 0
    if kind == 3:      # header bytes of the chunks: planes, bit counts, flags, mean, step
        idx = rng.choice(heads, size=rng.integers(1, 4))
    elif kind == 0:    # anywhere
        idx = rng.integers(0, n, size=rng.integers(1, 9))
    elif kind == 1:    # headers: the container header and the first chunk header behind it
        idx = rng.integers(0, min(n, 160), size=rng.integers(1, 5))
    else:              # cut short
        return bad[: rng.integers(0, n)].copy()
    bad[idx] = rng.integers(0, 256, size=len(idx), dtype=np.uint8)
    return bad


def run(trials=60, seed=0, eng=None):
    eng = eng or SperrHip()
    rng = np.random.default_rng(seed)
    vol = field((29, 33, 40), 1)
    img = field((1, 45, 52), 2)[0]
    cases = []

    def dec3(s):
        shape, _, _ = eng.parse_header(s)
        if shape[0] * shape[1] * shape[2] > 1 << 26:   # damaged dimensions: do not allocate that
            raise SperrHipError("dimensions out of range for this test")
        return eng.decompress(s, True, shape_zyx=shape)

    for mode, quality in ((1, 2.5), (2, 70.0), (3, 1e-3)):
        c = eng.compress(torch.from_numpy(vol).cuda(), (20, 20, 20), quality, mode=mode).cpu().numpy()
        cases.append((f"3d mode {mode}", c, dec3))
    cube = field((64, 64, 128), 3)   # chunks of a power-of-two size take the table-driven decoder
    for mode, quality in ((1, 2.0), (3, 1e-3)):
        c = eng.compress(torch.from_numpy(cube).cuda(), (64, 64, 64), quality, mode=mode).cpu().numpy()
        cases.append((f"3d 64-cube chunks mode {mode}", c, dec3))
    big = field((100, 90, 110), 4)   # one chunk whose lists mix set shapes and whose LIS phases span many regions of k_lis_mx
    for mode, quality in ((1, 3.0), (2, 80.0)):
        c = eng.compress(torch.from_numpy(big).cuda(), (110, 90, 100), quality, mode=mode).cpu().numpy()
        cases.append((f"3d 110x90x100 chunk mode {mode}", c, dec3))
    wide = field((1, 300, 420), 5)[0]
    c = eng.compress_2d(torch.from_numpy(wide).cuda(), 70.0, mode=2).cpu().numpy()
    cases.append(("2d 420x300 mode 2", c, lambda s: eng.decompress_2d(s, wide.shape, True)))
    for mode, quality in ((1, 3.0), (3, 1e-3)):
        c = eng.compress_2d(torch.from_numpy(img).cuda(), quality, mode=mode).cpu().numpy()
        cases.append((f"2d mode {mode}", c, lambda s: eng.decompress_2d(s, img.shape, True)))
    summary = []
    for name, good, dec in cases:
        ref = dec(torch.from_numpy(good).cuda()).cpu().numpy()
        rejected = decoded = 0
        try:
            heads = chunk_headers(good) if name.startswith("3d") else None
        except (IndexError, ValueError):
            heads = None
        for t in range(trials):
            bad = damage(rng, good, t % 4, heads)
            if len(bad) == 0:
                bad = good[:1].copy()
            try:
                out = dec(torch.from_numpy(bad).cuda())
                torch.cuda.synchronize()
                decoded += 1
                del out
            except SperrHipError:
                rejected += 1
        again = dec(torch.from_numpy(good).cuda()).cpu().numpy()
        same = np.array_equal(ref.view(np.uint32), again.view(np.uint32))
        summary.append((name, trials, rejected, decoded, same))
        print(f"{name}: {trials} damaged containers, {rejected} rejected, {decoded} decoded, "
              f"undamaged container afterwards {'identical' if same else 'DIFFERENT'}", flush=True)
    return summary


if __name__ == "__main__":
    res = run(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    sys.exit(0 if all(r[4] for r in res) else 1)

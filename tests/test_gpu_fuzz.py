"""Randomised parity sweep (-m gpu): random small volumes, chunkings, modes and qualities; the
HIP container must equal the oracle's byte for byte and decode to the same values."""
import os

import numpy as np
import pytest

from fields import smooth_field
from sperr_amd.api import SperrHipError
from sperr_amd.synth import turbulence

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU")
    from sperr_amd.api import SperrHip
    return SperrHip()


def cuda(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def bits(a):
    return a.view(np.uint64 if a.dtype == np.float64 else np.uint32)


def make_case(seed):
    rng = np.random.default_rng(seed)
    shape = tuple(int(v) for v in rng.integers(9, 56, size=3))            # (z, y, x)
    chunks = tuple(int(min(s, v)) for s, v in zip(shape[::-1], rng.integers(8, 60, size=3)))
    dtype = np.float32 if rng.random() < 0.7 else np.float64
    kind = rng.integers(0, 3)
    if kind == 0:
        v = turbulence(shape, seed=int(seed) + 1, dtype=dtype)
    elif kind == 1:
        v = smooth_field(shape, seed=int(seed) + 3, dtype=dtype)
    else:   # noisy: many outliers in PWE mode, several bit planes of them
        v = (turbulence(shape, seed=int(seed) + 5, dtype=np.float64) +
             0.2 * rng.standard_normal(shape) * (rng.random(shape) < 0.2)).astype(dtype)
    if rng.random() < 0.2:   # a constant block (constant chunk when the chunking lines up)
        v[: shape[0] // 2, : shape[1] // 2, : shape[2] // 2] = 1.5
    mode = int(rng.integers(1, 4))
    span = float(v.max() - v.min()) or 1.0
    if mode == 1:
        quality = float(rng.choice([0.3, 1.0, 2.5, 7.0, 19.0]))
    elif mode == 2:
        quality = float(rng.choice([30.0, 62.5, 101.0, 150.0]))
    else:
        quality = span * float(rng.choice([0.2, 3e-2, 1e-3, 1e-5]))
    return v, chunks, mode, quality


@pytest.mark.parametrize("seed", range(40))
def test_random_case(eng, seed):
    from oracle.pyoracle import Oracle
    oracle = Oracle()
    v, chunks, mode, quality = make_case(1000 + seed)
    want = oracle.comp_3d(v, chunks, mode, quality)
    got = bytes(eng.compress(cuda(v), chunks, quality, mode=mode).cpu().numpy())
    assert got == want, (v.shape, chunks, mode, quality, str(v.dtype))
    dev = cuda(np.frombuffer(want, dtype=np.uint8))
    for as_float in (True, False):
        assert np.array_equal(bits(eng.decompress(dev, as_float).cpu().numpy()),
                              bits(oracle.decomp_3d(want, as_float))), (v.shape, chunks, mode, quality)
    if mode == 3:
        assert np.abs(oracle.decomp_3d(want, False) - v.astype(np.float64)).max() <= quality


@pytest.mark.parametrize("seed", range(24))
def test_random_slice(eng, seed):
    """The 2D entry points on random slices: sperr_comp_2d / sperr_decomp_2d, the three modes."""
    from oracle.pyoracle import Oracle
    check_slice(eng, Oracle(), seed)


def make_slice(seed, maxdim=140):
    rng = np.random.default_rng(5000 + seed)
    shape = (int(rng.integers(9, maxdim)), int(rng.integers(9, maxdim)))
    dtype = np.float32 if rng.random() < 0.6 else np.float64
    img = turbulence((1,) + shape, seed=seed + 11, dtype=dtype)[0]
    if rng.random() < 0.3:
        img = (img + 0.1 * rng.standard_normal(shape)).astype(dtype)
    mode = int(rng.integers(1, 4))
    span = float(img.max() - img.min()) or 1.0
    quality = (float(rng.choice([0.4, 1.5, 4.0, 11.0])), float(rng.choice([35.0, 70.0, 120.0, 200.0])),
               span * float(rng.choice([0.1, 1e-2, 1e-4, 1e-7])))[mode - 1]
    hdr = bool(rng.integers(0, 2))
    return rng, shape, dtype, img, mode, quality, hdr


def check_slice(eng, oracle, seed, maxdim=140):
    rng, shape, dtype, img, mode, quality, hdr = make_slice(seed, maxdim)
    want = oracle.comp_2d(img, mode, quality, hdr)
    got = bytes(eng.compress_2d(cuda(img), quality, mode=mode, header=hdr).cpu().numpy())
    assert got == want, (shape, mode, quality, str(dtype))
    body = want[10:] if hdr else want
    cut = len(body) if rng.random() < 0.5 else max(27, int(len(body) * rng.random()))
    dev = cuda(np.frombuffer(body[:cut], dtype=np.uint8))
    for as_float in (True, False):
        assert np.array_equal(bits(eng.decompress_2d(dev, shape, as_float).cpu().numpy()),
                              bits(oracle.decomp_2d(body[:cut], shape, as_float))), (shape, mode, quality, cut)


def test_damaged_containers_are_rejected_or_decoded(eng):
    """Bytes flipped anywhere, inside the container / chunk / outlier headers, or the container cut
    short: the decoder returns an error or some values, and the undamaged container still decodes to
    exactly what it did before (tools/fuzz_corrupt.py runs more of these)."""
    import importlib.util
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools", "fuzz_corrupt.py")
    spec = importlib.util.spec_from_file_location("fuzz_corrupt", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for name, trials, rejected, decoded, same in mod.run(trials=48, seed=7, eng=eng):
        assert same, name
        assert rejected + decoded == trials


def test_header_with_absurd_dimensions_is_rejected(eng):
    """A header naming more chunks than the container has bytes must be refused before anything is
    sized from it (no allocation failure escapes the C boundary)."""
    v = turbulence((24, 24, 24), seed=3)
    good = eng.compress(cuda(v), (12, 12, 12), 2.0).cpu().numpy()
    for vol, chunk in (((0x7fffffff,) * 3, (1, 1, 1)), ((0xffffffff,) * 3, (2, 3, 1)),
                       ((1 << 22,) * 3, (0xffff,) * 3)):
        bad = good.copy()
        bad[2:14] = np.frombuffer(np.array(vol, dtype=np.uint32).tobytes(), dtype=np.uint8)
        bad[14:20] = np.frombuffer(np.array(chunk, dtype=np.uint16).tobytes(), dtype=np.uint8)
        with pytest.raises(SperrHipError):
            eng.parse_header(cuda(bad))
        with pytest.raises(SperrHipError):
            eng.decompress(cuda(bad), True, shape_zyx=(24, 24, 24))
        with pytest.raises(SperrHipError):
            eng.decomp_3d(bytes(bad))
        with pytest.raises(SperrHipError):
            eng.trunc_3d(bytes(bad), 50)


def test_bit_counts_near_2_to_64_do_not_wrap(eng):
    """A PWE chunk whose outlier header claims 2^64 - k bits (k < 8) with nothing behind it: rounding
    the bit count up to bytes must not wrap to zero (the 9-byte tail would pass for a complete outlier
    stream and every buffer sized from it would be tiny).  The same for the SPECK header's count.
    The decoder may reject the container or decode it without outliers; afterwards the undamaged
    container gives what it gave before."""
    v = (turbulence((24, 24, 24), seed=9, dtype=np.float64) +
         0.3 * (np.random.default_rng(2).random((24, 24, 24)) < 0.1)).astype(np.float32)
    for chunks, nchunk in (((24, 24, 24), 1), ((24, 24, 8), 3)):     # one chunk; eight-divisible batches aside
        good = eng.compress(cuda(v), chunks, 1e-3, mode=3).cpu().numpy()
        ref = eng.decompress(cuda(good), True).cpu().numpy()
        hdr = (14 if nchunk == 1 else 20) + 4 * nchunk
        lens = np.frombuffer(good[hdr - 4 * nchunk:hdr].tobytes(), dtype=np.uint32).astype(np.int64)
        assert hdr + lens.sum() == len(good)
        tb = int(np.frombuffer(good[hdr + 18:hdr + 26].tobytes(), dtype=np.uint64)[0])
        speck_end = hdr + 26 + (tb + 7) // 8          # first chunk: outlier header starts here
        assert speck_end + 9 < hdr + lens[0], "first chunk has no outlier stream; pick a noisier field"
        for k in (1, 3, 7, 8, 9):
            # chunk 0 cut right behind its outlier header, which now claims 2^64 - k bits
            bad = np.concatenate([good[:speck_end + 9], good[hdr + lens[0]:]]).copy()
            bad[speck_end + 1:speck_end + 9] = np.frombuffer(np.uint64((1 << 64) - k).tobytes(), dtype=np.uint8)
            newlen = np.uint32(speck_end + 9 - hdr)
            bad[hdr - 4 * nchunk:hdr - 4 * nchunk + 4] = np.frombuffer(newlen.tobytes(), dtype=np.uint8)
            try:
                eng.decompress(cuda(bad), True, shape_zyx=(24, 24, 24))
            except SperrHipError:
                pass
            # the SPECK stream's own count
            bad2 = good.copy()
            bad2[hdr + 18:hdr + 26] = np.frombuffer(np.uint64((1 << 64) - k).tobytes(), dtype=np.uint8)
            try:
                eng.decompress(cuda(bad2), True, shape_zyx=(24, 24, 24))
            except SperrHipError:
                pass
        again = eng.decompress(cuda(good), True).cpu().numpy()
        assert np.array_equal(ref.view(np.uint32), again.view(np.uint32))

"""CPU: the oracle restatement against the REAL reference (oracle/_ref, compiled from
/root/reference by oracle/Makefile), stage by stage and end to end, bit for bit.  Skipped where
oracle/_ref has not been built."""
import numpy as np
import pytest

from fields import ramp_field, smooth_field
from sperr_amd.synth import turbulence

SHAPES = [(17, 17, 17), (32, 32, 32), (23, 45, 70), (41, 64, 64), (9, 40, 48), (64, 64, 64)]


def bits(a):
    return a.view(np.uint64 if a.dtype == np.float64 else np.uint32)


@pytest.mark.parametrize("shape", SHAPES)
def test_dwt_idwt_conditioner_bit_exact(oracle, ref, shape):
    v = turbulence(shape).astype(np.float64)
    a, b = oracle.dwt3d(v), ref.dwt3d(v)
    assert np.array_equal(bits(a), bits(b))
    assert np.array_equal(bits(oracle.idwt3d(a)), bits(ref.idwt3d(b)))
    ca, ha, _ = oracle.condition(v)
    cb, hb, _ = ref.condition(v)
    assert ha == hb and np.array_equal(bits(ca), bits(cb))


@pytest.mark.parametrize("shape", SHAPES[:5])
@pytest.mark.parametrize("budget", [0, 4096, 100000])
def test_speck_streams_bit_exact(oracle, ref, shape, budget):
    v = oracle.dwt3d(turbulence(shape).astype(np.float64))
    q = np.abs(v).max() / 60000.0
    coef, sign, width = oracle.quantize(v, q)
    so = oracle.speck3d_encode(coef, sign, budget)
    sr = ref.speck3d_encode(coef, sign, budget, width=8)
    assert so == sr
    co, sgo = oracle.speck3d_decode(sr, shape)
    cr, sgr = ref.speck3d_decode(sr, shape)
    assert np.array_equal(co, cr)
    nz = cr.reshape(-1) != 0
    idx = np.nonzero(nz)[0]
    bo = (sgo[idx >> 6] >> (idx & 63).astype(np.uint64)) & np.uint64(1)
    br = (sgr[idx >> 6] >> (idx & 63).astype(np.uint64)) & np.uint64(1)
    assert np.array_equal(bo, br)
    # truncated stream (progressive access): same prefix decodes identically
    cut = 9 + (len(sr) - 9) // 3
    co, _ = oracle.speck3d_decode(sr[:cut], shape)
    cr, _ = ref.speck3d_decode(sr[:cut], shape)
    assert np.array_equal(co, cr)


@pytest.mark.parametrize("chunks", [(64, 64, 64), (32, 32, 32), (40, 30, 20)])
@pytest.mark.parametrize("bpp", [0.5, 2.0, 6.5])
def test_container_bit_exact(oracle, ref, chunks, bpp):
    v = turbulence((50, 64, 72))
    so, sr = oracle.comp_3d(v, chunks, 1, bpp), ref.comp_3d(v, chunks, 1, bpp)
    assert so == sr
    assert np.array_equal(bits(oracle.decomp_3d(sr, True)), bits(ref.decomp_3d(sr, True)))
    assert np.array_equal(bits(oracle.decomp_3d(sr, False)), bits(ref.decomp_3d(sr, False)))


def test_high_precision_retry_bit_exact(oracle, ref):
    """src/SPECK_FLT.cpp:530-538: too few bits at 32 planes -> re-quantise for 53 planes."""
    r = ramp_field((16, 16, 16))
    for bpp in (30.0, 60.0):
        so, sr = oracle.comp_3d(r, (16, 16, 16), 1, bpp), ref.comp_3d(r, (16, 16, 16), 1, bpp)
        assert sr[18 + 17] == 53 and so == sr
        assert np.array_equal(bits(oracle.decomp_3d(sr)), bits(ref.decomp_3d(sr)))


def test_chunk_volume_matches(oracle, ref):
    for vol, ch in [((128, 128, 41), (64, 64, 41)), ((91, 91, 91), (64, 64, 64)),
                    ((100, 70, 33), (30, 40, 8)), ((5, 5, 5), (9, 9, 9))]:
        ch = tuple(min(c, v) for c, v in zip(ch, vol))
        assert np.array_equal(oracle.chunk_volume(vol, ch), ref.chunk_volume(vol, ch))


def test_double_input_and_f64_field(oracle, ref):
    v = smooth_field((24, 40, 40), dtype=np.float64)
    so, sr = oracle.comp_3d(v, (40, 40, 24), 1, 3.0), ref.comp_3d(v, (40, 40, 24), 1, 3.0)
    assert so == sr and not (sr[1] & 0x20)


@pytest.mark.parametrize("chunks", [(32, 32, 32), (48, 40, 24)])
@pytest.mark.parametrize("psnr", [40.0, 85.0, 140.0, 230.0])
def test_psnr_mode_bit_exact(oracle, ref, chunks, psnr):
    """Mode 2 (src/SPECK_FLT.cpp:237-279,431-452): q search, integer width, full-depth coding."""
    v = turbulence((48, 40, 64))
    want = ref.comp_3d(v, chunks, 2, psnr)
    assert oracle.comp_3d(v, chunks, 2, psnr) == want
    assert np.array_equal(bits(oracle.decomp_3d(want, False)), bits(ref.decomp_3d(want, False)))
    d = smooth_field((24, 40, 40), dtype=np.float64)
    assert oracle.comp_3d(d, chunks, 2, psnr) == ref.comp_3d(d, chunks, 2, psnr)


@pytest.mark.parametrize("chunks", [(32, 32, 32), (48, 40, 24), (64, 40, 48)])
@pytest.mark.parametrize("tol", [0.3, 1e-2, 1e-4, 1e-9])
def test_pwe_mode_bit_exact(oracle, ref, chunks, tol):
    """Mode 3 (src/SPECK_FLT.cpp:280-281,461-486,573-584): q = 1.5 tol, full-depth coding, and the
    outlier list through Outlier_Coder / SPECK1D_INT; decoded values honour the tolerance."""
    v = turbulence((48, 40, 64))
    want = ref.comp_3d(v, chunks, 3, tol)
    assert oracle.comp_3d(v, chunks, 3, tol) == want
    dec = oracle.decomp_3d(want, False)
    assert np.array_equal(bits(dec), bits(ref.decomp_3d(want, False)))
    assert np.abs(dec - v.astype(np.float64)).max() <= tol
    d = smooth_field((24, 40, 40), dtype=np.float64) * 1e-3
    want = ref.comp_3d(d, chunks, 3, tol)
    assert oracle.comp_3d(d, chunks, 3, tol) == want
    assert np.array_equal(bits(oracle.decomp_3d(want, False)), bits(ref.decomp_3d(want, False)))


@pytest.mark.parametrize("shape,chunks", [((64, 64, 64), (32, 32, 32)), ((48, 64, 32), (32, 32, 24)),
                                          ((40, 40, 40), (40, 40, 40)), ((41, 64, 64), (64, 64, 41))])
def test_multi_resolution_decode_bit_exact(oracle, ref, shape, chunks):
    """SPERR3D_OMP_D::decompress(p, true): volume and hierarchy (src/SPERR3D_OMP_D.cpp:50-150,
    src/CDF97.cpp:150-168, src/SPECK_FLT.cpp:592-603) against the reference's own classes."""
    v = turbulence(shape)
    stream = ref.comp_3d(v, chunks, 1, 3.0)
    vol_o, lv_o = oracle.decomp_3d_multi_res(stream)
    vol_r, lv_r = ref.decomp_3d_multi_res(stream)
    assert np.array_equal(bits(vol_o), bits(vol_r))
    assert [a.shape for a in lv_o] == [a.shape for a in lv_r]
    for a, b in zip(lv_o, lv_r):
        assert np.array_equal(bits(a), bits(b))


@pytest.mark.parametrize("shape", [(64, 64), (37, 50), (96, 121), (9, 200), (150, 11)])
@pytest.mark.parametrize("mode,quality", [(1, 0.7), (1, 5.0), (2, 75.0), (2, 160.0), (3, 1e-2), (3, 1e-6)])
def test_2d_slices_bit_exact(oracle, ref, shape, mode, quality):
    """sperr_comp_2d / sperr_decomp_2d (src/SPERR_C_API.cpp:7-134): dwt2d, SPECK2D_INT with its
    type-I set (src/SPECK2D_INT*.cpp), all three modes, with and without the 10-byte header."""
    for dtype in (np.float32, np.float64):
        img = turbulence((1,) + shape, dtype=dtype)[0]
        for hdr in (False, True):
            want = ref.comp_2d(img, mode, quality, hdr)
            assert oracle.comp_2d(img, mode, quality, hdr) == want
        body = want[10:]
        for as_float in (True, False):
            assert np.array_equal(bits(oracle.decomp_2d(body, shape, as_float)),
                                  bits(ref.decomp_2d(body, shape, as_float)))


@pytest.mark.parametrize("shape", [(64, 64), (37, 50), (96, 121), (9, 200), (150, 11), (7, 7), (100, 128)])
@pytest.mark.parametrize("mode,quality", [(1, 3.0), (2, 90.0), (3, 1e-3)])
def test_2d_multi_resolution_bit_exact(oracle, ref, shape, mode, quality):
    """SPECK2D_FLT::decompress(multi_res = true) (src/SPECK2D_FLT.cpp:52-58, src/CDF97.cpp:114-130,
    src/SPECK_FLT.cpp:592-603): the slice and the slice at every coarsened resolution, full and
    truncated streams."""
    img = turbulence((1,) + shape, dtype=np.float64 if mode == 3 else np.float32)[0]
    stream = ref.comp_2d(img, mode, quality, False)
    for cut in (len(stream), max(27, len(stream) * 2 // 5)):
        a, la = oracle.decomp_2d_multi_res(stream[:cut], shape)
        b, lb = ref.decomp_2d_multi_res(stream[:cut], shape)
        assert np.array_equal(bits(a), bits(b))
        assert np.array_equal(bits(a), bits(ref.decomp_2d(stream[:cut], shape, False)))
        assert [x.shape for x in la] == [x.shape for x in lb]
        for x, y in zip(la, lb):
            assert np.array_equal(bits(x), bits(y))


def test_speck1d_bit_exact(oracle, ref):
    """SPECK1D_INT_ENC / _DEC (src/SPECK1D_INT*.cpp) on sparse arrays: same stream, and it
    round-trips exactly through both decoders."""
    rng = np.random.default_rng(5)
    for n, k, top in [(3, 2, 3), (5, 5, 9), (100, 7, 200), (4097, 300, 5), (65536, 1000, 70000)]:
        coef = np.zeros(n, dtype=np.uint64)
        pos = rng.choice(n, size=min(k, n), replace=False)
        coef[pos] = rng.integers(1, top + 1, size=pos.size, dtype=np.uint64)
        sign = rng.integers(0, 2, size=n).astype(bool)
        stream = oracle.speck1d_encode(coef, sign)
        assert stream == ref.speck1d_encode(coef, sign)
        for dec in (oracle, ref):
            c2, s2 = dec.speck1d_decode(stream, n)
            assert np.array_equal(c2, coef)
            assert np.array_equal(s2[coef > 0], sign[coef > 0])


@pytest.mark.parametrize("chunks", [(64, 40, 48), (32, 32, 32), (20, 18, 16)])
@pytest.mark.parametrize("pct", [0, 1, 10, 37, 75, 100, 150])
def test_progressive_truncation_bit_exact(oracle, ref, chunks, pct):
    """sperr_trunc_3d (src/SPERR3D_Stream_Tools.cpp:134-226): same bytes, and the truncated
    container decodes to the same values."""
    v = turbulence((48, 40, 64))
    v[:16, :18, :20] = 2.5   # a constant chunk for the (20, 18, 16) chunking: 17-byte stream
    full = ref.comp_3d(v, chunks, 1, 3.0)
    want = ref.trunc_3d(full, pct)
    assert oracle.trunc_3d(full, pct) == want
    assert np.array_equal(bits(oracle.decomp_3d(want, True)), bits(ref.decomp_3d(want, True)))


def test_integer_len_rule_matches_the_reference(oracle, ref):
    """SPECK_FLT::integer_len() (src/SPECK_FLT.cpp:193-213) of the reference's encoder and decoder
    against the rule include/sperr_hip.hpp derives from the oracle's stream: byte 17 of a chunk
    stream, the number of bit planes, <= 8 / 16 / 32 / more -> 1 / 2 / 4 / 8 bytes."""
    import ctypes as C
    corner = turbulence((32, 32, 32)).astype(np.float64)
    seen = set()
    for psnr in (20.0, 60.0, 120.0, 250.0):
        w = (C.c_size_t * 2)()
        ref.probe.refp_integer_len_psnr.restype = C.c_int
        ref.probe.refp_integer_len_psnr.argtypes = [C.c_void_p] + [C.c_size_t] * 3 + [C.c_double, C.c_void_p]
        assert ref.probe.refp_integer_len_psnr(corner.ctypes.data, 32, 32, 32, psnr, w) == 0
        nbp = oracle.comp_3d(corner, (32, 32, 32), 2, psnr)[18 + 17]
        rule = 1 if nbp <= 8 else 2 if nbp <= 16 else 4 if nbp <= 32 else 8
        assert w[0] == w[1] == rule
        seen.add(rule)
    assert seen == {1, 2, 4, 8}

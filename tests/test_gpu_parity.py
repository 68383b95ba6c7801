"""GPU parity tests (-m gpu): the HIP path, called through the C ABI of libsperr_hip.so, against
the oracle (oracle/sperr_oracle.c, pinned to the reference) and the committed golden vectors.

Bar: bit-exact containers and SPECK streams (integer / byte work); decoded floats are compared
bit-for-bit as well -- the stated tolerance is 1 ULP of the output type, and the fp64 pipeline is
reproduced exactly, so the observed difference is 0."""
import hashlib
import json
import os

import numpy as np
import pytest

from fields import ramp_field, smooth_field
from sperr_amd.synth import turbulence

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


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


def ulp_diff_f32(a, b):
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7fffffff), ia)
    ib = np.where(ib < 0, -(ib & 0x7fffffff), ib)
    return int(np.abs(ia - ib).max())


DWT_SHAPES = [(17, 17, 17), (32, 32, 32), (23, 45, 70), (41, 64, 64), (9, 40, 48), (64, 64, 64),
              (128, 128, 128), (20, 300, 9)]


@pytest.mark.parametrize("shape", DWT_SHAPES)
def test_dwt_idwt_bit_exact(eng, oracle, shape):
    v = turbulence(shape).astype(np.float64)
    want = oracle.dwt3d(v)
    d = cuda(v)
    eng.dwt3d(d)
    got = d.cpu().numpy()
    assert np.array_equal(bits(got), bits(want))
    eng.dwt3d(d, inverse=True)
    assert np.array_equal(bits(d.cpu().numpy()), bits(oracle.idwt3d(want)))


def quantized(oracle, shape, scale):
    v = oracle.dwt3d(turbulence(shape).astype(np.float64))
    q = np.abs(v).max() / scale
    coef, sign, _ = oracle.quantize(v, q)
    return coef, sign


def to_dev_coef(coef, wide):
    if wide:
        return cuda(coef.astype(np.uint64).view(np.int64))
    return cuda(coef.astype(np.uint32).view(np.int32))


SPECK_SHAPES = [(8, 8, 8), (16, 16, 16), (17, 17, 17), (13, 21, 30), (32, 32, 32), (9, 40, 48),
                (41, 64, 64), (3, 5, 7), (1, 16, 16), (2, 2, 2), (48, 48, 48), (64, 64, 64)]


@pytest.mark.parametrize("shape", SPECK_SHAPES)
@pytest.mark.parametrize("budget", [0, 1000, 20000])
def test_speck_encode_bit_exact(eng, oracle, shape, budget):
    coef, sign = quantized(oracle, shape, 3000.0)
    want = oracle.speck3d_encode(coef, sign, budget)
    got = eng.speck3d_encode(to_dev_coef(coef, False), cuda(sign.view(np.int64)), budget)
    assert got[:9] == want[:9]
    assert got == want


@pytest.mark.parametrize("shape", [(16, 16, 16), (13, 21, 30), (32, 32, 32)])
def test_speck_encode_wide_and_32bit_range(eng, oracle, shape):
    for scale, wide in [(4294967295.0, False), (float(2 ** 53 - 1), True)]:
        coef, sign = quantized(oracle, shape, scale)
        for budget in (0, 50000):
            want = oracle.speck3d_encode(coef, sign, budget)
            got = eng.speck3d_encode(to_dev_coef(coef, wide), cuda(sign.view(np.int64)), budget)
            assert got == want


def test_speck_encode_zero_and_sparse(eng, oracle):
    coef = np.zeros((12, 12, 12), dtype=np.uint64)
    sign = np.full((coef.size + 63) // 64, np.uint64(0xFFFFFFFFFFFFFFFF))
    for _ in range(3):
        want = oracle.speck3d_encode(coef, sign, 0)
        got = eng.speck3d_encode(to_dev_coef(coef, False), cuda(sign.view(np.int64)), 0)
        assert got == want
        coef[3, 4, 5] += 77
        coef[11, 0, 2] += 1


@pytest.mark.parametrize("shape", SPECK_SHAPES)
@pytest.mark.parametrize("budget", [0, 1000, 20000])
def test_speck_decode_bit_exact(eng, oracle, shape, budget):
    coef, sign = quantized(oracle, shape, 3000.0)
    stream = oracle.speck3d_encode(coef, sign, budget)
    for cut in (len(stream), 9 + (len(stream) - 9) // 2, 9 + (len(stream) - 9) // 7):
        s = stream[:cut]
        c0, s0 = oracle.speck3d_decode(s, shape)
        c1, s1 = eng.speck3d_decode(s, shape)
        assert np.array_equal(c0, c1)
        assert np.array_equal(s0, s1)


def test_speck_decode_wide(eng, oracle):
    shape = (16, 20, 24)
    coef, sign = quantized(oracle, shape, float(2 ** 53 - 1))
    stream = oracle.speck3d_encode(coef, sign, 0)
    assert stream[0] > 32
    c0, s0 = oracle.speck3d_decode(stream, shape)
    c1, s1 = eng.speck3d_decode(stream, shape)
    assert np.array_equal(c0, c1) and np.array_equal(s0, s1)


with open(os.path.join(GOLD, "golden.json")) as f:
    _G = json.load(f)
    CASES, CASES_2D = _G["cases"], _G["cases_2d"]


@pytest.mark.parametrize("case", CASES, ids=[c["tag"] for c in CASES])
def test_golden_vectors(eng, case):
    """Containers produced by the REAL reference (tests/golden/make_golden.py)."""
    name, shape = case["input"], tuple(case["shape_zyx"])
    path = os.path.join(GOLD, name + ".f32")
    if os.path.exists(path):
        arr = np.fromfile(path, dtype=np.float32).reshape(shape)
    else:
        arr = smooth_field(shape, dtype=np.dtype(case["dtype"]))
    assert hashlib.sha256(arr.tobytes()).hexdigest() == case["input_sha256"]
    with open(os.path.join(GOLD, case["tag"] + ".sperr"), "rb") as f:
        want = f.read()
    got = bytes(eng.compress(cuda(arr), case["chunks_xyz"], case["bpp"],
                             mode=case.get("mode", 1)).cpu().numpy())
    assert len(got) == case["stream_len"]
    assert got == want
    dev = cuda(np.frombuffer(want, dtype=np.uint8))
    dec_f = eng.decompress(dev, True).cpu().numpy()
    dec_d = eng.decompress(dev, False).cpu().numpy()
    assert hashlib.sha256(dec_f.tobytes()).hexdigest() == case["decoded_f32_sha256"]
    assert hashlib.sha256(dec_d.tobytes()).hexdigest() == case["decoded_f64_sha256"]


@pytest.mark.parametrize("shape,chunks", [((50, 64, 72), (32, 32, 32)), ((50, 64, 72), (40, 30, 20)),
                                          ((64, 64, 64), (64, 64, 64)), ((41, 128, 128), (64, 64, 41)),
                                          ((96, 96, 96), (48, 48, 48))])
@pytest.mark.parametrize("bpp", [0.5, 2.0, 6.5])
def test_container_matches_oracle(eng, oracle, shape, chunks, bpp):
    v = turbulence(shape)
    want = oracle.comp_3d(v, chunks, 1, bpp)
    got = bytes(eng.compress(cuda(v), chunks, bpp).cpu().numpy())
    assert got == want
    dev = cuda(np.frombuffer(want, dtype=np.uint8))
    f_got, f_want = eng.decompress(dev, True).cpu().numpy(), oracle.decomp_3d(want, True)
    assert ulp_diff_f32(f_got, f_want) <= 1           # stated tolerance: 1 ULP (fp32)
    assert np.array_equal(bits(f_got), bits(f_want))  # observed: identical
    assert np.array_equal(bits(eng.decompress(dev, False).cpu().numpy()),
                          bits(oracle.decomp_3d(want, False)))


@pytest.mark.parametrize("shape,chunks", [((50, 64, 72), (32, 32, 32)), ((64, 64, 64), (64, 64, 64)),
                                          ((41, 128, 128), (64, 64, 41)), ((96, 96, 96), (48, 48, 48))])
@pytest.mark.parametrize("psnr", [55.0, 90.0, 130.0])
def test_psnr_mode_matches_oracle(eng, oracle, shape, chunks, psnr):
    """Mode 2 (src/SPECK_FLT.cpp:237-279): the q search and the full-depth coding give the
    reference's bytes; the target itself is met."""
    v = turbulence(shape)
    want = oracle.comp_3d(v, chunks, 2, psnr)
    got = bytes(eng.compress(cuda(v), chunks, psnr, mode=2).cpu().numpy())
    assert got == want
    dev = cuda(np.frombuffer(want, dtype=np.uint8))
    f_got, f_want = eng.decompress(dev, False).cpu().numpy(), oracle.decomp_3d(want, False)
    assert np.array_equal(bits(f_got), bits(f_want))
    rng = float(v.max()) - float(v.min())
    mse = float(np.mean((f_got - v.astype(np.float64)) ** 2))
    assert 10.0 * np.log10(rng * rng / mse) >= psnr - 0.01


def test_psnr_mode_wide_coefficients_and_c_api(eng, oracle):
    """A PSNR so high that the coefficients need more than 32 bits (SPECK_FLT.cpp:324-337), a
    volume with a constant chunk, double input, and the host C API in mode 2."""
    v = smooth_field((32, 32, 64), dtype=np.float64)
    v[:, :, :32] = 0.75
    want = oracle.comp_3d(v, (32, 32, 32), 2, 230.0)
    assert want[20 + 8 + 17 + 17] > 32                     # second chunk: more than 32 bit planes
    got = bytes(eng.compress(cuda(v), (32, 32, 32), 230.0, mode=2).cpu().numpy())
    assert got == want
    assert eng.comp_3d(v, (32, 32, 32), 2, 230.0) == want
    dev = cuda(np.frombuffer(want, dtype=np.uint8))
    assert np.array_equal(bits(eng.decompress(dev, False).cpu().numpy()),
                          bits(oracle.decomp_3d(want, False)))


@pytest.mark.parametrize("shape,chunks", [((50, 64, 72), (32, 32, 32)), ((64, 64, 64), (64, 64, 64)),
                                          ((41, 128, 128), (64, 64, 41)), ((96, 96, 96), (48, 48, 48))])
@pytest.mark.parametrize("tol", [0.2, 1e-2, 1e-4])
def test_pwe_mode_matches_oracle(eng, oracle, shape, chunks, tol):
    """Mode 3 (src/SPECK_FLT.cpp:280-281,461-486,573-584): q = 1.5 tol, full-depth coding, the
    outlier list through the 1D coder (src/Outlier_Coder.cpp, src/SPECK1D_INT*.cpp): the
    reference's bytes, its decoded values, and the tolerance itself is met."""
    v = turbulence(shape)
    want = oracle.comp_3d(v, chunks, 3, tol)
    got = bytes(eng.compress(cuda(v), chunks, tol, mode=3).cpu().numpy())
    assert got == want
    dev = cuda(np.frombuffer(want, dtype=np.uint8))
    f_got, f_want = eng.decompress(dev, False).cpu().numpy(), oracle.decomp_3d(want, False)
    assert np.array_equal(bits(f_got), bits(f_want))
    assert np.abs(f_got - v.astype(np.float64)).max() <= tol


def test_pwe_mode_wide_coefficients_constant_chunk_and_c_api(eng, oracle):
    """A tolerance so small that the coefficients need more than 32 bits, a volume with a constant
    chunk, double input, and the host C API in mode 3."""
    v = smooth_field((32, 32, 64), dtype=np.float64)
    v[:, :, :32] = 0.75
    for tol in (1e-9, 1e-2):
        want = oracle.comp_3d(v, (32, 32, 32), 3, tol)
        got = bytes(eng.compress(cuda(v), (32, 32, 32), tol, mode=3).cpu().numpy())
        assert got == want
        assert eng.comp_3d(v, (32, 32, 32), 3, tol) == want
        dev = cuda(np.frombuffer(want, dtype=np.uint8))
        f = eng.decompress(dev, False).cpu().numpy()
        assert np.array_equal(bits(f), bits(oracle.decomp_3d(want, False)))
        assert np.abs(f - v).max() <= tol


def test_pwe_mode_many_outliers(eng, oracle):
    """Noise on top of a smooth field: a few per cent of the values become outliers, on several
    bit planes of the 1D coder."""
    rng = np.random.default_rng(11)
    v = (turbulence((64, 64, 64)) + 0.3 * rng.standard_normal((64, 64, 64))).astype(np.float32)
    for tol in (0.05, 0.5):
        want = oracle.comp_3d(v, (64, 64, 64), 3, tol)
        got = bytes(eng.compress(cuda(v), (64, 64, 64), tol, mode=3).cpu().numpy())
        assert got == want
        dev = cuda(np.frombuffer(want, dtype=np.uint8))
        assert np.array_equal(bits(eng.decompress(dev, True).cpu().numpy()),
                              bits(oracle.decomp_3d(want, True)))


def test_high_precision_retry(eng, oracle):
    """src/SPECK_FLT.cpp:530-538: 32 planes cannot fill the budget -> 53 planes, 64-bit ints."""
    r = ramp_field((16, 16, 16))
    for bpp in (30.0, 60.0):
        want = oracle.comp_3d(r, (16, 16, 16), 1, bpp)
        assert want[18 + 17] == 53
        got = bytes(eng.compress(cuda(r), (16, 16, 16), bpp).cpu().numpy())
        assert got == want
        dev = cuda(np.frombuffer(want, dtype=np.uint8))
        assert np.array_equal(bits(eng.decompress(dev, True).cpu().numpy()),
                              bits(oracle.decomp_3d(want, True)))
    # a volume where only SOME chunks retry
    v = turbulence((32, 32, 64))
    v[:, :, 32:] = ramp_field((32, 32, 32))
    want = oracle.comp_3d(v, (32, 32, 32), 1, 40.0)
    assert bytes(eng.compress(cuda(v), (32, 32, 32), 40.0).cpu().numpy()) == want
    dev = cuda(np.frombuffer(want, dtype=np.uint8))
    assert np.array_equal(bits(eng.decompress(dev, False).cpu().numpy()),
                          bits(oracle.decomp_3d(want, False)))


@pytest.mark.parametrize("bpp", [2.0, 9.0, 14.0, 16.0, 17.0, 18.0, 18.5, 19.0, 19.5, 20.0, 24.0, 29.5])
def test_decoded_coefficients_hand_over_schemes(eng, oracle, bpp):
    """k_ref_assemble hands the inverse passes the sign inside the coefficient word where the chunk allows it
    (coef_scheme, speck_dec.h): fixed-rate chunks have 32 planes (src/SPECK_FLT.cpp:282-290), so which scheme a chunk
    takes depends on the plane its stream runs out on -- from plane 2 up the low bit makes room, below it the
    magnitudes stay as they are and the masks are read.  A sweep of rates down to the last planes, two chunk shapes
    (the fused x-y-z inverse kernel and the per-axis passes)."""
    for shape in ((32, 32, 32), (16, 40, 24)):
        v = turbulence(shape)
        want = oracle.comp_3d(v, shape, 1, bpp)
        assert want[18 + 17] >= 32   # (more than 32: the 64-bit retry took over, SPECK_FLT.cpp:530-538)
        dev = cuda(np.frombuffer(want, dtype=np.uint8))
        assert np.array_equal(bits(eng.decompress(dev, True).cpu().numpy()), bits(oracle.decomp_3d(want, True)))
        assert np.array_equal(bits(eng.decompress(dev, False).cpu().numpy()), bits(oracle.decomp_3d(want, False)))


def test_retry_of_a_batch_whose_coder_arrays_lay_over_the_chunk_buffer(eng, oracle):
    """Round 3: the coder's node arrays, birth records and second list lie over the fp64 chunk buffer once
    the quantiser has read it (carve_enc); a batch that then needs the 64-bit retry (src/SPECK_FLT.cpp:530-538)
    transforms its chunks again and moves those arrays to scratch memory (wide_retry_prepare).  64^3 chunks
    are large enough for the overlay (smaller ones lose it to the 256-byte rounding of ten arrays): one
    chunk, then a volume whose two shape groups run side by side and retry after both are enqueued."""
    import ctypes as C
    eng.lib.sperrhip_debug_counter.restype = C.c_ulonglong
    eng.lib.sperrhip_debug_counter.argtypes = [C.c_int]
    over0, redo0 = eng.lib.sperrhip_debug_counter(1), eng.lib.sperrhip_debug_counter(0)
    r = ramp_field((64, 64, 64))
    want = oracle.comp_3d(r, (64, 64, 64), 1, 40.0)
    assert want[18 + 17] == 53
    assert bytes(eng.compress(cuda(r), (64, 64, 64), 40.0).cpu().numpy()) == want
    assert eng.lib.sperrhip_debug_counter(1) > over0, "the coder arrays were not laid over the chunk buffer"
    assert eng.lib.sperrhip_debug_counter(0) > redo0, "the retry did not transform the batch again"
    dev = cuda(np.frombuffer(want, dtype=np.uint8))
    assert np.array_equal(bits(eng.decompress(dev, True).cpu().numpy()), bits(oracle.decomp_3d(want, True)))
    # two shape groups (64^3 and 64 x 64 x 70) side by side, some chunks retry; rate, PSNR and PWE mode
    v = turbulence((64, 64, 134))
    v[:, :, :64] = ramp_field((64, 64, 64))
    # (mode 2 at 200 dB: both chunks need 35 planes, so the retry runs with wantRange -- the conditioner's
    # range atomics again -- while q from psnr_q_search has to survive the rewritten coder state)
    for mode, q in ((1, 40.0), (2, 200.0), (3, 1e-9)):
        redo1 = eng.lib.sperrhip_debug_counter(0)
        want = oracle.comp_3d(v, (64, 64, 64), mode, q)
        if mode == 2:
            assert want[28 + 17] > 32, "the PSNR case no longer needs the 64-bit pass"
        assert bytes(eng.compress(cuda(v), (64, 64, 64), q, mode=mode).cpu().numpy()) == want, mode
        if mode != 3:   # (round 5: in point-wise error mode the coder's arrays have memory of their own -- its outlier
            #            stage writes the chunk buffer while the coder runs --, so its retry transforms nothing again)
            assert eng.lib.sperrhip_debug_counter(0) > redo1, mode
        dev = cuda(np.frombuffer(want, dtype=np.uint8))
        assert np.array_equal(bits(eng.decompress(dev, False).cpu().numpy()), bits(oracle.decomp_3d(want, False)))


@pytest.mark.parametrize("shape,chunks,mode,q", [((32, 64, 64), (16, 16, 16), 1, 3.0), ((48, 32, 48), (16, 16, 24), 3, 1e-3),
                                                 ((16, 48, 64), (16, 16, 16), 2, 70.0), ((20, 33, 47), (10, 11, 12), 1, 6.0)])
def test_small_batches_decode_in_several_sub_batches(eng, oracle, shape, chunks, mode, q):
    """Round 3: a decompression call that has the device to itself cuts a batch of 4 to 55 equally shaped
    chunks into two or four sub-batches that decode side by side (decompress_impl: streams, events and
    workspace per sub-batch, outlier streams included); 32, 12, 12 and 24 chunks here, all three modes."""
    v = turbulence(shape)
    want = oracle.comp_3d(v, chunks[::-1], mode, q)
    assert bytes(eng.compress(cuda(v), chunks[::-1], q, mode=mode).cpu().numpy()) == want
    dev = cuda(np.frombuffer(want, dtype=np.uint8))
    for as_float in (True, False):
        assert np.array_equal(bits(eng.decompress(dev, as_float).cpu().numpy()), bits(oracle.decomp_3d(want, as_float)))


def test_compact_chunk_buffer_of_the_decoder(eng, oracle):
    """Round 3: when the finest level is the fused x-y-z kernel and every inverse pass dequantises on load,
    the decoder's fp64 chunk buffer holds the second level's box only (carve_dec, compact_box): dyadic
    chunks of even and odd extents, several per batch, against the oracle."""
    import ctypes as C
    eng.lib.sperrhip_debug_counter.restype = C.c_ulonglong
    eng.lib.sperrhip_debug_counter.argtypes = [C.c_int]
    for shape, chunks in (((64, 64, 128), (64, 64, 64)), ((33, 45, 82), (33, 45, 41)), ((40, 48, 56), (20, 24, 56))):
        c0 = eng.lib.sperrhip_debug_counter(2)
        v = turbulence(shape)
        want = oracle.comp_3d(v, chunks[::-1], 1, 3.0)
        dev = cuda(np.frombuffer(want, dtype=np.uint8))
        assert np.array_equal(bits(eng.decompress(dev, True).cpu().numpy()), bits(oracle.decomp_3d(want, True))), shape
        assert np.array_equal(bits(eng.decompress(dev, False).cpu().numpy()), bits(oracle.decomp_3d(want, False))), shape
        assert eng.lib.sperrhip_debug_counter(2) > c0, ("compact buffer not used", shape)


def test_constant_and_mixed_chunks(eng, oracle):
    v = turbulence((32, 32, 64))
    v[:, :, :32] = 1.25                       # first chunk constant -> 17-byte stream
    want = oracle.comp_3d(v, (32, 32, 32), 1, 2.0)
    got = bytes(eng.compress(cuda(v), (32, 32, 32), 2.0).cpu().numpy())
    assert got == want
    dev = cuda(np.frombuffer(want, dtype=np.uint8))
    assert np.array_equal(bits(eng.decompress(dev, True).cpu().numpy()),
                          bits(oracle.decomp_3d(want, True)))
    c = np.full((16, 20, 32), -3.5, dtype=np.float32)
    assert bytes(eng.compress(cuda(c), (32, 20, 16), 2.0).cpu().numpy()) == \
        oracle.comp_3d(c, (32, 20, 16), 1, 2.0)


def test_double_input(eng, oracle):
    v = smooth_field((24, 40, 40), dtype=np.float64)
    want = oracle.comp_3d(v, (40, 40, 24), 1, 3.0)
    assert bytes(eng.compress(cuda(v), (40, 40, 24), 3.0).cpu().numpy()) == want


def test_reference_c_api_drop_in(eng, oracle):
    """sperr_comp_3d / sperr_decomp_3d / sperr_parse_header with host buffers and the reference's
    return codes (include/SPERR_C_API.h:100-105)."""
    import ctypes as C
    v = turbulence((40, 48, 56))
    want = oracle.comp_3d(v, (32, 32, 32), 1, 2.0)
    got = eng.comp_3d(v, (32, 32, 32), 1, 2.0)
    assert got == want
    assert np.array_equal(bits(eng.decomp_3d(want, True)), bits(oracle.decomp_3d(want, True)))
    assert np.array_equal(bits(eng.decomp_3d(want, False)), bits(oracle.decomp_3d(want, False)))
    dx, dy, dz, isf = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0), C.c_int(0)
    buf = np.frombuffer(want, dtype=np.uint8)
    eng.lib.sperr_parse_header(buf.ctypes.data, C.byref(dx), C.byref(dy), C.byref(dz), C.byref(isf))
    assert (dx.value, dy.value, dz.value, isf.value) == (56, 48, 40, 1)
    # return codes: *dst not NULL -> 1 ; quality <= 0 -> 2 ; unknown mode -> 2
    dst, n = C.c_void_p(1234), C.c_size_t(0)
    args = (v.ctypes.data, 1, 56, 48, 40, 32, 32, 32)
    assert eng.lib.sperr_comp_3d(*args, 1, 2.0, 0, C.byref(dst), C.byref(n)) == 1
    dst = C.c_void_p(None)
    assert eng.lib.sperr_comp_3d(*args, 1, -1.0, 0, C.byref(dst), C.byref(n)) == 2
    assert eng.lib.sperr_comp_3d(*args, 7, 2.0, 0, C.byref(dst), C.byref(n)) == 2


@pytest.mark.parametrize("pct", [5, 40, 100])
def test_progressive_truncation(eng, oracle, pct):
    """sperr_trunc_3d: same bytes as the oracle's restatement of the reference, and the portion
    decodes on the GPU to the same values (missing bits read as zero)."""
    v = turbulence((64, 64, 96))
    full = oracle.comp_3d(v, (32, 32, 32), 1, 4.0)
    want = oracle.trunc_3d(full, pct)
    assert eng.trunc_3d(full, pct) == want
    assert (want[1] & 0x80) == (0x80 if pct < 100 else 0)
    dev = cuda(np.frombuffer(want, dtype=np.uint8))
    assert np.array_equal(bits(eng.decompress(dev, True).cpu().numpy()),
                          bits(oracle.decomp_3d(want, True)))
    assert np.array_equal(bits(eng.decomp_3d(want, False)), bits(oracle.decomp_3d(want, False)))


def test_truncated_container_is_rejected(eng, oracle):
    v = turbulence((32, 32, 32))
    s = oracle.comp_3d(v, (32, 32, 32), 1, 2.0)
    import ctypes as C
    buf = np.frombuffer(s[:-5], dtype=np.uint8)
    dst = C.c_void_p(None)
    d = [C.c_size_t(0) for _ in range(3)]
    assert eng.lib.sperr_decomp_3d(buf.ctypes.data, buf.size, 1, 0, C.byref(d[0]), C.byref(d[1]),
                                   C.byref(d[2]), C.byref(dst)) == -1


def test_256_cube_chunk_bpp2(eng, oracle):
    """The metric's unit of work: one 256^3 fp32 chunk at 2 bpp, byte-identical to the oracle."""
    v = turbulence((256, 256, 256))
    want = oracle.comp_3d(v, (256, 256, 256), 1, 2.0)
    got = bytes(eng.compress(cuda(v), (256, 256, 256), 2.0).cpu().numpy())
    assert len(got) == 4194348
    assert got == want
    dev = cuda(np.frombuffer(want, dtype=np.uint8))
    assert np.array_equal(bits(eng.decompress(dev, True).cpu().numpy()),
                          bits(oracle.decomp_3d(want, True)))


def test_config2_fp64_128cube_chunks_pwe(eng, oracle):
    """BASELINE.json configs[1]: a 128x128x256 fp64 field in 128^3 chunks at PWE = 1e-6 (the named
    file is not in the reference tree; SURVEY.md 8(d) substitutes the synthetic field, kept in
    fp64).  Same container as the oracle, same decoded doubles, tolerance met."""
    v = turbulence((256, 128, 128), dtype=np.float64)
    want = oracle.comp_3d(v, (128, 128, 128), 3, 1e-6, nthreads=2)
    got = bytes(eng.compress(cuda(v), (128, 128, 128), 1e-6, mode=3).cpu().numpy())
    assert got == want
    dev = cuda(np.frombuffer(want, dtype=np.uint8))
    f = eng.decompress(dev, False).cpu().numpy()
    assert np.array_equal(bits(f), bits(oracle.decomp_3d(want, False)))
    assert np.abs(f - v).max() <= 1e-6


def test_pwe_full_size_properties(eng):
    """512^3 fp32 in 256^3 chunks, point-wise error mode: the tolerance holds for every value,
    two runs give the same bytes, and a chunk's stream does not depend on its neighbours."""
    import struct
    import torch
    from sperr_amd.synth import turbulence_torch
    vol = turbulence_torch((512, 512, 512), "cuda")
    tol = 2e-3
    s1 = eng.compress(vol, (256, 256, 256), tol, mode=3).clone()
    s2 = eng.compress(vol, (256, 256, 256), tol, mode=3)
    assert torch.equal(s1, s2)
    back = eng.decompress(s1, True)
    # (the float cast of the output may add half an ulp of the value to the fp64 error)
    assert (back.double() - vol.double()).abs().max().item() <= tol * (1 + 1e-3)
    lens = struct.unpack_from("<8I", bytes(s1[20:52].cpu().numpy()))
    sub = vol[:256, :256, 256:].contiguous()
    s_sub = eng.compress(sub, (256, 256, 256), tol, mode=3)
    hdr = 20 + 4 * 8
    assert s_sub.numel() == 18 + lens[1]
    assert torch.equal(s_sub[18:], s1[hdr + lens[0]: hdr + lens[0] + lens[1]])


@pytest.mark.parametrize("shape,chunks", [((64, 64, 64), (32, 32, 32)), ((48, 64, 32), (32, 32, 24)),
                                          ((64, 64, 64), (64, 64, 64)), ((40, 40, 40), (40, 40, 40)),
                                          ((128, 128, 128), (128, 128, 128))])
def test_multi_resolution_decode(eng, oracle, shape, chunks):
    """SPERR3D_OMP_D::decompress(p, multi_res = true) (src/SPERR3D_OMP_D.cpp:50-150,
    src/CDF97.cpp:150-168): the volume and every coarsened level, bit for bit."""
    v = turbulence(shape)
    v[: shape[0] // 2, : chunks[1], : chunks[0]] = 0.25   # a constant chunk when the grid is 2x2x2
    stream = oracle.comp_3d(v, chunks, 1, 3.0)
    want_vol, want_levels = oracle.decomp_3d_multi_res(stream)
    assert eng.multires_levels(shape, chunks) == [lv.shape for lv in want_levels]
    dev = cuda(np.frombuffer(stream, dtype=np.uint8))
    vol, levels = eng.decompress_multires(dev, output_float=False)
    assert np.array_equal(bits(vol.cpu().numpy()), bits(want_vol))
    assert len(levels) == len(want_levels) > 0
    for got, want in zip(levels, want_levels):
        assert np.array_equal(bits(got.cpu().numpy()), bits(want))


@pytest.mark.parametrize("tol", [2e-2, 1e-4])
def test_pwe_container_through_both_inverse_paths(eng, oracle, tol):
    """A point-wise error container written by the encoder's FUSED reconstruction (pwe_stage_begin: the decoder's
    values are rebuilt with k_lift_xyz_inv before the outliers are found, src/SPECK_FLT.cpp:461-486) decodes within the
    tolerance, and to the reference's bits, through BOTH inverse paths of the decoder: the fused brick (plain
    decompression) and the per-axis k_lift_axis passes (the resolution hierarchy keeps every pass in the chunk buffer).
    The guarantee rests on the two paths computing the same fp64 values; this pins it for PWE containers."""
    shape, chunks = (64, 128, 128), (64, 64, 64)
    v = turbulence(shape)
    want = oracle.comp_3d(v, chunks, 3, tol)
    got = bytes(eng.compress(cuda(v), chunks, tol, mode=3).cpu().numpy())
    assert got == want
    dev = cuda(np.frombuffer(want, dtype=np.uint8))
    ref = oracle.decomp_3d(want, False)
    plain = eng.decompress(dev, False).cpu().numpy()
    vol, levels = eng.decompress_multires(dev, output_float=False)
    assert len(levels) > 0
    for back in (plain, vol.cpu().numpy()):
        assert np.array_equal(bits(back), bits(ref))
        assert np.abs(back - v.astype(np.float64)).max() <= tol


def test_multi_resolution_absent_for_other_shapes(eng, oracle):
    """Chunks that are not dyadic (wavelet-packet transform), or do not tile the volume, have no
    hierarchy (src/sperr_helper.cpp:70-123); the volume still decodes."""
    for shape, chunks in [((9, 64, 64), (64, 64, 9)), ((50, 64, 72), (32, 32, 32))]:
        v = turbulence(shape)
        stream = oracle.comp_3d(v, chunks, 1, 2.0)
        assert oracle.decomp_3d_multi_res(stream)[1] == []
        assert eng.multires_levels(shape, chunks) == []
        dev = cuda(np.frombuffer(stream, dtype=np.uint8))
        vol, levels = eng.decompress_multires(dev, output_float=True)
        assert levels == []
        assert np.array_equal(bits(vol.cpu().numpy()), bits(oracle.decomp_3d(stream, True)))


@pytest.mark.parametrize("case", CASES_2D, ids=[c["tag"] for c in CASES_2D])
def test_golden_vectors_2d(eng, case):
    """sperr_comp_2d / sperr_decomp_2d against streams written by the reference itself."""
    shape = tuple(case["shape_yx"])
    path = os.path.join(GOLD, case["input"] + ".f32")
    if os.path.exists(path):
        arr = np.fromfile(path, dtype=np.float32).reshape(shape)
    else:
        arr = smooth_field((1,) + shape, dtype=np.dtype(case["dtype"]))[0]
    with open(os.path.join(GOLD, case["tag"] + ".sperr"), "rb") as f:
        want = f.read()
    got = bytes(eng.compress_2d(cuda(arr), case["quality"], mode=case["mode"],
                                header=case["header"]).cpu().numpy())
    assert got == want
    body = cuda(np.frombuffer(want[10:] if case["header"] else want, dtype=np.uint8))
    dec_f = eng.decompress_2d(body, shape, True).cpu().numpy()
    dec_d = eng.decompress_2d(body, shape, False).cpu().numpy()
    assert hashlib.sha256(dec_f.tobytes()).hexdigest() == case["decoded_f32_sha256"]
    assert hashlib.sha256(dec_d.tobytes()).hexdigest() == case["decoded_f64_sha256"]


@pytest.mark.parametrize("shape", [(64, 64), (37, 50), (96, 121), (9, 200), (150, 11), (256, 256)])
@pytest.mark.parametrize("mode,quality", [(1, 0.7), (1, 5.0), (2, 75.0), (2, 160.0), (3, 1e-2), (3, 1e-6)])
def test_2d_slices_match_oracle(eng, oracle, shape, mode, quality):
    """dwt2d + SPECK2D_INT with its type-I set, all three modes (src/SPECK2D_INT*.cpp,
    src/SPERR_C_API.cpp:7-134), float and double input, with and without the header."""
    for dtype in (np.float32, np.float64):
        img = turbulence((1,) + shape, dtype=dtype)[0]
        for hdr in (False, True):
            want = oracle.comp_2d(img, mode, quality, hdr)
            got = bytes(eng.compress_2d(cuda(img), quality, mode=mode, header=hdr).cpu().numpy())
            assert got == want
        body = cuda(np.frombuffer(want[10:], dtype=np.uint8))
        for as_float in (True, False):
            assert np.array_equal(bits(eng.decompress_2d(body, shape, as_float).cpu().numpy()),
                                  bits(oracle.decomp_2d(want[10:], shape, as_float)))


def test_2d_host_c_api_and_truncated_stream(eng, oracle):
    """The host entry points (malloc'd results, return codes) and a stream cut short: the decoder
    pads with zeros like the reference (src/SPECK_INT.cpp:95-105)."""
    import ctypes as C
    img = turbulence((1, 80, 120))[0]
    want = oracle.comp_2d(img, 1, 3.0, False)
    lib = eng.lib
    dst, n = C.c_void_p(None), C.c_size_t(0)
    lib.sperr_comp_2d.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.c_int, C.c_double,
                                  C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    assert lib.sperr_comp_2d(img.ctypes.data, 1, 120, 80, 1, 3.0, 0, C.byref(dst), C.byref(n)) == 0
    assert C.string_at(dst.value, n.value) == want
    assert lib.sperr_comp_2d(img.ctypes.data, 1, 120, 80, 1, 3.0, 0, C.byref(dst), C.byref(n)) == 1
    C.CDLL(None).free(dst)
    dst = C.c_void_p(None)
    assert lib.sperr_comp_2d(img.ctypes.data, 1, 120, 80, 7, 3.0, 0, C.byref(dst), C.byref(n)) == 2
    assert lib.sperr_comp_2d(img.ctypes.data, 1, 120, 80, 1, -1.0, 0, C.byref(dst), C.byref(n)) == 2
    for cut in (len(want), len(want) - 700, 60):
        part = want[:cut]
        buf = np.frombuffer(part, dtype=np.uint8)
        out = C.c_void_p(None)
        lib.sperr_decomp_2d.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_size_t, C.c_size_t,
                                        C.POINTER(C.c_void_p)]
        assert lib.sperr_decomp_2d(buf.ctypes.data, buf.size, 0, 120, 80, C.byref(out)) == 0
        got = np.frombuffer(C.string_at(out.value, 80 * 120 * 8), dtype=np.float64).reshape(80, 120)
        C.CDLL(None).free(out)
        assert np.array_equal(bits(got.copy()), bits(oracle.decomp_2d(part, (80, 120), False)))


@pytest.mark.parametrize("shape", [(64, 64), (37, 50), (9, 200), (150, 11), (7, 7), (200, 333), (1024, 256)])
@pytest.mark.parametrize("mode,quality", [(1, 3.0), (2, 90.0), (3, 1e-3)])
def test_2d_multi_resolution_decode(eng, oracle, shape, mode, quality):
    """SPECK2D_FLT::decompress(multi_res = true): the slice at every coarsened resolution, coarsest
    first, full and truncated streams, float and double slices; the level shapes; a constant slice."""
    img = turbulence((1,) + shape, dtype=np.float64 if mode == 3 else np.float32)[0]
    stream = oracle.comp_2d(img, mode, quality, False)
    for cut in (len(stream), max(27, len(stream) * 2 // 5)):
        want, want_lv = oracle.decomp_2d_multi_res(stream[:cut], shape)
        dev = cuda(np.frombuffer(stream[:cut], dtype=np.uint8))
        assert eng.multires_levels_2d(shape) == [x.shape for x in want_lv]
        for as_float in (True, False):
            got, got_lv = eng.decompress_2d_multires(dev, shape, as_float)
            ref_vol = want.astype(np.float32) if as_float else want
            assert np.array_equal(bits(got.cpu().numpy()), bits(ref_vol))
            assert len(got_lv) == len(want_lv)
            for a, b in zip(got_lv, want_lv):
                assert np.array_equal(bits(a.cpu().numpy()), bits(b))
    flat = np.full(shape, 2.5, dtype=np.float32)
    s = oracle.comp_2d(flat, 1, 2.0, False)
    got, got_lv = eng.decompress_2d_multires(cuda(np.frombuffer(s, dtype=np.uint8)), shape, False)
    assert np.all(got.cpu().numpy() == 2.5) and all(np.all(a.cpu().numpy() == 2.5) for a in got_lv)


def test_2d_multi_resolution_host_api(eng, oracle):
    import ctypes as C
    shape = (90, 141)
    img = turbulence((1,) + shape)[0]
    stream = oracle.comp_2d(img, 1, 4.0, False)
    want, want_lv = oracle.decomp_2d_multi_res(stream, shape)
    buf = np.frombuffer(stream, dtype=np.uint8)
    dst, nlev = C.c_void_p(None), C.c_size_t(0)
    ldims, levels = (C.c_size_t * 32)(), (C.c_void_p * 16)()
    f = eng.lib.sperrhip_decomp_2d_multires
    f.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_size_t, C.c_size_t, C.POINTER(C.c_void_p),
                  C.POINTER(C.c_size_t), C.c_void_p, C.c_void_p]
    assert f(buf.ctypes.data, buf.size, 0, shape[1], shape[0], C.byref(dst), C.byref(nlev), ldims, levels) == 0
    assert nlev.value == len(want_lv)
    got = np.frombuffer(C.string_at(dst.value, want.size * 8), dtype=np.float64).reshape(shape)
    assert np.array_equal(bits(got.copy()), bits(want))
    for h, lv in enumerate(want_lv):
        assert (ldims[2 * h + 1], ldims[2 * h]) == lv.shape
        a = np.frombuffer(C.string_at(levels[h], lv.size * 8), dtype=np.float64).reshape(lv.shape)
        assert np.array_equal(bits(a.copy()), bits(lv))
        C.CDLL(None).free(C.c_void_p(levels[h]))
    assert f(buf.ctypes.data, buf.size, 0, shape[1], shape[0], C.byref(dst), C.byref(nlev), ldims, levels) == 1
    C.CDLL(None).free(dst)


@pytest.mark.parametrize("shape,chunks,dtype", [((10, 37, 300), (300, 37, 10), np.float32),
                                                ((9, 520, 543), (543, 520, 9), np.float32),
                                                ((12, 100, 511), (511, 100, 12), np.float64),
                                                ((16, 33, 400), (200, 33, 8), np.float32),
                                                ((24, 258, 258), (258, 258, 24), np.float32)])
def test_long_rows_and_odd_tiles(eng, oracle, shape, chunks, dtype):
    """Rows longer than 256 samples: the fused x/y lifting kernel (k_lift_xy) takes the rows of a
    tile in several rounds and its last tile is ragged; mirrored halos at every border."""
    v = turbulence(shape, dtype=dtype)
    want = oracle.comp_3d(v, chunks, 1, 4.0)
    assert bytes(eng.compress(cuda(v), chunks, 4.0).cpu().numpy()) == want
    dev = cuda(np.frombuffer(want, dtype=np.uint8))
    for as_float in (True, False):
        assert np.array_equal(bits(eng.decompress(dev, as_float).cpu().numpy()),
                              bits(oracle.decomp_3d(want, as_float)))


def test_many_chunks_decode_in_sub_batches(eng, oracle):
    """48 chunks of one shape: the decoder splits the batch into sub-batches on separate streams
    (engine.hip, decompress_impl); same values as the oracle."""
    v = turbulence((192, 256, 256))
    want = oracle.comp_3d(v, (64, 64, 64), 1, 2.0, nthreads=8)
    got = eng.compress(cuda(v), (64, 64, 64), 2.0)
    assert bytes(got.cpu().numpy()) == want
    assert np.array_equal(bits(eng.decompress(got, True).cpu().numpy()),
                          bits(oracle.decomp_3d(want, True)))


def test_full_size_roundtrip_properties(eng):
    """512^3 in 256^3 chunks at 2 bpp: size-independent properties -- exact stream length,
    determinism (checksum of two runs), bounded error, chunk independence (a chunk's stream does
    not depend on its neighbours)."""
    import torch
    from sperr_amd.synth import turbulence_torch
    vol = turbulence_torch((512, 512, 512), "cuda")
    s1 = eng.compress(vol, (256, 256, 256), 2.0).clone()
    s2 = eng.compress(vol, (256, 256, 256), 2.0)
    assert s1.numel() == 20 + 4 * 8 + 8 * (17 + 9 + 4194304)
    assert torch.equal(s1, s2)
    back = eng.decompress(s1, True)
    err = (back.double() - vol.double()).abs().max().item()
    rng = (vol.max() - vol.min()).item()
    assert err < 5e-3 * rng
    sub = vol[:256, :256, 256:].contiguous()
    s_sub = eng.compress(sub, (256, 256, 256), 2.0)
    hdr = 20 + 4 * 8
    one = 17 + 9 + 4194304
    assert torch.equal(s_sub[18:], s1[hdr + one: hdr + 2 * one])


def test_sixteen_256_cube_chunks_reach_the_capped_grids(eng, oracle):
    """1024 x 512 x 512 in 256^3 chunks at 2 bpp, byte for byte against the oracle: sixteen chunks of one shape
    are what it takes for the decoder's per-plane sweeps to run on their CAPPED grids (from 16 chunks on: 4096
    workgroups over the batch, launch_speck_decode -- 1024 pixel tiles per chunk, so every workgroup strides over
    four tiles) and for the encoder's workgroups to stride over ten to twenty-five tiles each (1536 / 768
    workgroups over the batch since round 6, launch_speck_encode_planes); the suite's other many-chunk cases use
    16^3 chunks, far under the caps, and until round 6 only the bench's 1024^3 comparison got here.  The decoder cuts sixteen chunks into four sub-batches of four when it has the device to itself
    (wide grids again), so the container is decoded once more in a fresh process with SPERR_HIP_SUBSTREAMS=1: one
    batch of sixteen.  The field's period is 300 samples, so no two chunks hold the same data (the default
    period is the chunk size).  A tile loop with a broken stride fails this case (tried on MI355X with
    `tile += gridDim.x + 1` in k_list_count -- the container differs -- and in k_ref_deposit -- the child's
    values differ).  The chunk loop of the reference: src/SPERR3D_OMP_C.cpp:94-130."""
    import subprocess
    import sys
    import tempfile
    from sperr_amd.synth import turbulence_torch
    vol = turbulence_torch((1024, 512, 512), "cuda", seed=3, period=300.0)
    host = vol.cpu().numpy()
    want = oracle.comp_3d(host, (256, 256, 256), 1, 2.0)
    del host
    got = eng.compress(vol, (256, 256, 256), 2.0)
    assert bytes(got.cpu().numpy()) == want
    del vol
    ref = oracle.decomp_3d(want, True)
    assert np.array_equal(bits(eng.decompress(got, True).cpu().numpy()), bits(ref))
    with tempfile.TemporaryDirectory() as td:
        np.save(os.path.join(td, "c.npy"), np.frombuffer(want, dtype=np.uint8))
        np.save(os.path.join(td, "r.npy"), ref)
        del ref
        code = ("import sys, numpy as np, torch; sys.path.insert(0, %r); from sperr_amd.api import SperrHip; "
                "e = SperrHip(); c = torch.from_numpy(np.load(%r)).cuda(); r = np.load(%r); "
                "d = e.decompress(c, True).cpu().numpy(); "
                "sys.exit(0 if np.array_equal(d.view(np.uint32), r.view(np.uint32)) else 3)"
                % (os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."), os.path.join(td, "c.npy"),
                   os.path.join(td, "r.npy")))
        env = dict(os.environ, SPERR_HIP_SUBSTREAMS="1")
        assert subprocess.run([sys.executable, "-c", code], env=env, timeout=300).returncode == 0


def test_streams_that_end_anywhere_in_the_list_kernels(eng, oracle):
    """One 128^3 chunk (4096 sets of 8^3: k_lis_l2 takes their list from 512 entries on, round 6) at twenty-four rates
    between 0.25 and 6 bits per sample: the stream's end falls inside the LIP scan, inside the list of the 2^3, the 4^3 or
    the 8^3 sets (k_lis_l0 / _l1 / _l2), inside k_lis_hi's lists or inside a refinement pass, on planes whose lists
    are short and on planes whose lists are long -- every decoded value against the reference's bits
    (src/SPECK_INT.cpp:165-228: the decoder stops the moment the bits run out; src/SPECK3D_INT.cpp:99-212)."""
    v = turbulence((128, 128, 128), seed=11)
    for i in range(24):
        bpp = 0.25 + 0.25 * i
        want = oracle.comp_3d(v, (128, 128, 128), 1, bpp)
        dev = cuda(np.frombuffer(want, dtype=np.uint8))
        got = eng.decompress(dev, True).cpu().numpy()
        assert np.array_equal(bits(got), bits(oracle.decomp_3d(want, True))), bpp


def test_256_cube_chunk_pwe(eng, oracle):
    """One 256^3 fp32 chunk in point-wise error mode (the chunk shape of BASELINE configs 3 and 5 with
    config 5's mode): outlier stream included, byte-identical to the oracle; decoded floats
    bit-identical; tolerance met."""
    v = turbulence((256, 256, 256))
    tol = 1e-3
    want = oracle.comp_3d(v, (256, 256, 256), 3, tol)
    got = bytes(eng.compress(cuda(v), (256, 256, 256), tol, mode=3).cpu().numpy())
    assert got == want
    dev = cuda(np.frombuffer(want, dtype=np.uint8))
    f = eng.decompress(dev, True).cpu().numpy()
    assert np.array_equal(bits(f), bits(oracle.decomp_3d(want, True)))
    assert np.abs(f.astype(np.float64) - v.astype(np.float64)).max() <= tol * (1 + 1e-6) + 1e-7


def test_config1_128_cube_single_chunk_bpp4(eng, oracle):
    """BASELINE.json configs[0] (wmag128.float is not in the reference tree: SURVEY 8(d) substitutes
    the synthetic 128^3 field): one chunk at BPP = 4.0."""
    v = turbulence((128, 128, 128))
    want = oracle.comp_3d(v, (128, 128, 128), 1, 4.0)
    assert len(want) == 18 + 26 + 128 ** 3 * 4 // 8
    got = bytes(eng.compress(cuda(v), (128, 128, 128), 4.0).cpu().numpy())
    assert got == want
    dev = cuda(np.frombuffer(want, dtype=np.uint8))
    assert np.array_equal(bits(eng.decompress(dev, True).cpu().numpy()), bits(oracle.decomp_3d(want, True)))


def test_more_than_65535_chunks(eng, oracle):
    """The reference cuts a volume into any number of chunks (src/sperr_helper.cpp:542-592); a grid
    dimension of the device is limited to 65535, which the container kernels must not depend on."""
    v = turbulence((168, 168, 168))
    chunks = (4, 4, 4)                                # 42^3 = 74088 chunks
    want = oracle.comp_3d(v, chunks, 1, 8.0, nthreads=8)
    got = bytes(eng.compress(cuda(v), chunks, 8.0).cpu().numpy())
    assert got == want
    dev = cuda(np.frombuffer(want, dtype=np.uint8))
    assert np.array_equal(bits(eng.decompress(dev, True).cpu().numpy()), bits(oracle.decomp_3d(want, True)))


def test_output_buffer_smaller_than_the_header_is_refused(eng):
    """sperrhip_compress_dev takes the room of d_dst: a buffer that cannot even hold the container
    header must be refused before any kernel writes to it."""
    import ctypes as C
    import torch
    v = cuda(turbulence((16, 16, 16)))
    small = torch.zeros(64, dtype=torch.uint8, device="cuda")
    n = C.c_size_t(0)
    for cap in (0, 5, 13, 19, 30):
        rtn = eng.lib.sperrhip_compress_dev(v.data_ptr(), 1, 16, 16, 16, 8, 8, 8, 1, 2.0, small.data_ptr(), cap,
                                            C.byref(n), None)
        assert rtn == -1
    torch.cuda.synchronize()
    assert int(small[32:].sum().item()) == 0


@pytest.mark.parametrize("shape", [(1, 1, 3), (1, 1, 2), (1, 2, 1), (1, 1, 1), (2, 2, 2), (1, 1, 5), (3, 1, 1)])
def test_tiny_chunks_pwe(eng, oracle, shape):
    """Chunks of a few samples in point-wise error mode: the reference accepts them (no transform,
    q = 1.5 tol, never an outlier); containers and decoded doubles as the oracle's."""
    v = (np.arange(int(np.prod(shape)), dtype=np.float64).reshape(shape) * 0.37 + 0.1) ** 2
    for tol in (1e-1, 1e-3, 1e-9):
        want = oracle.comp_3d(v, shape[::-1], 3, tol)
        assert bytes(eng.compress(cuda(v), shape[::-1], tol, mode=3).cpu().numpy()) == want
        back = eng.decompress(cuda(np.frombuffer(want, dtype=np.uint8)), False).cpu().numpy()
        assert np.array_equal(bits(back), bits(oracle.decomp_3d(want, False)))


@pytest.mark.parametrize("shape", [(16, 16, 1024), (1, 1024, 1024), (32, 32, 1024), (8, 8, 4096)])
def test_long_class_chains(eng, oracle, shape):
    """Chunks with an axis of 1024 samples: list levels whose class chain is 9 long (sets of 512 down
    to 2 along x), which the GPU-wide list kernel (k_lis_hi) takes (its predecessor k_lis_tables did not); an
    axis of 4096 (chains of 11) goes to k_lis_mx, whose rows are keyed by shape class.
    Streams, truncated streams and whole containers against the oracle."""
    coef, sign = quantized(oracle, shape, 200000.0)
    stream = oracle.speck3d_encode(coef, sign, 0)
    assert eng.speck3d_encode(to_dev_coef(coef, False), cuda(sign.view(np.int64)), 0) == stream
    for cut in (len(stream), 9 + (len(stream) - 9) // 3):
        c0, s0 = oracle.speck3d_decode(stream[:cut], shape)
        c1, s1 = eng.speck3d_decode(stream[:cut], shape)
        assert np.array_equal(c0, c1) and np.array_equal(s0, s1)
    v = turbulence(shape)
    want = oracle.comp_3d(v, shape[::-1], 1, 3.0)
    assert bytes(eng.compress(cuda(v), shape[::-1], 3.0).cpu().numpy()) == want
    back = eng.decompress(cuda(np.frombuffer(want, dtype=np.uint8)), True).cpu().numpy()
    assert np.array_equal(bits(back), bits(oracle.decomp_3d(want, True)))


@pytest.mark.parametrize("shape,chunks,bpp", [((129, 129, 129), (129, 129, 129), 2.0),
                                              ((33, 70, 100), (30, 40, 8), 3.0),
                                              ((96, 96, 192), (96, 96, 96), 1.0),
                                              ((80, 80, 80), (80, 80, 80), 4.0),
                                              ((100, 100, 100), (100, 100, 100), 0.5),
                                              ((17, 300, 21), (17, 300, 21), 2.0)])
def test_mixed_shape_chunks(eng, oracle, shape, chunks, bpp):
    """Chunks whose lists mix set shapes (k_lis_mx: shape-class rows and one walking wavefront per chunk,
    /root/reference/src/SPECK3D_INT.cpp:214-326 is the split rule behind the classes): odd lengths,
    lengths of the form 3 * 2^k, levels that hold leaf parents and larger sets side by side,
    wavelet-packet shapes; containers and decoded values identical to the oracle, also for a
    stream cut short."""
    v = turbulence(shape)
    want = oracle.comp_3d(v, chunks, 1, bpp)
    got = bytes(eng.compress(cuda(v), chunks, bpp).cpu().numpy())
    assert got == want
    dev = cuda(np.frombuffer(want, dtype=np.uint8))
    assert np.array_equal(bits(eng.decompress(dev, False).cpu().numpy()), bits(oracle.decomp_3d(want, False)))
    part = oracle.trunc_3d(want, 37)
    assert np.array_equal(bits(eng.decompress(cuda(np.frombuffer(part, dtype=np.uint8)), True).cpu().numpy()),
                          bits(oracle.decomp_3d(part, True)))


def test_250_cube_chunk(eng, oracle):
    """One 250^3 chunk at 2 bpp (decoded through a serial walk in 1.4 s): container and
    decoded floats identical to the oracle's."""
    v = turbulence((250, 250, 250))
    want = oracle.comp_3d(v, (250, 250, 250), 1, 2.0)
    got = bytes(eng.compress(cuda(v), (250, 250, 250), 2.0).cpu().numpy())
    assert got == want
    dev = cuda(np.frombuffer(want, dtype=np.uint8))
    assert np.array_equal(bits(eng.decompress(dev, True).cpu().numpy()), bits(oracle.decomp_3d(want, True)))


def test_block_of_a_1000_cube_volume(eng, oracle):
    """A chunk of 232 x 256 x 256 cut from a 1000^3 field (what `chunk_volume` leaves at the border of
    such a volume in 256^3 chunks).  Regression: in one of its planes a list of sets that are walked
    into holds a run of more than 64 insignificant entries right after a window restart; the
    walker's class window (k_lis_mixed then, k_lis_mx now) was taken for loaded there and one entry decoded as a leaf
    parent -- every later bit of the chunk was read out of place."""
    from sperr_amd.synth import turbulence_torch
    v = turbulence_torch((1000, 1000, 1000), "cuda", seed=7)[256:512, 256:512, 768:1000].contiguous()
    hv = v.cpu().numpy()
    want = oracle.comp_3d(hv, (232, 256, 256), 1, 2.0)
    assert bytes(eng.compress(v, (232, 256, 256), 2.0).cpu().numpy()) == want
    back = eng.decompress(cuda(np.frombuffer(want, dtype=np.uint8)), True).cpu().numpy()
    assert np.array_equal(bits(back), bits(oracle.decomp_3d(want, True)))


def test_serial_walk_fallback_in_a_fresh_process(oracle):
    """`SPERR_HIP_LIS_MIXED=0` (read once per process) sends chunks whose lists mix set shapes through the
    serial walk `k_lis_walk`, which also is what trees the class machinery does not take fall back to:
    the same bits as the oracle's."""
    import subprocess
    import sys
    import tempfile
    shape = (21, 34, 27)
    v = turbulence(shape)
    want = oracle.comp_3d(v, shape[::-1], 1, 3.0)
    ref = oracle.decomp_3d(want, True)
    with tempfile.TemporaryDirectory() as td:
        np.save(os.path.join(td, "c.npy"), np.frombuffer(want, dtype=np.uint8))
        np.save(os.path.join(td, "r.npy"), ref)
        code = ("import sys, numpy as np, torch; sys.path.insert(0, %r); from sperr_amd.api import SperrHip; "
                "e = SperrHip(); c = torch.from_numpy(np.load(%r)).cuda(); r = np.load(%r); "
                "d = e.decompress(c, True).cpu().numpy(); "
                "sys.exit(0 if np.array_equal(d.view(np.uint32), r.view(np.uint32)) else 3)"
                % (os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."), os.path.join(td, "c.npy"),
                   os.path.join(td, "r.npy")))
        env = dict(os.environ, SPERR_HIP_LIS_MIXED="0")
        assert subprocess.run([sys.executable, "-c", code], env=env, timeout=300).returncode == 0


@pytest.mark.parametrize("knobs", [{"SPERR_HIP_LIS_HI": "0"}, {"SPERR_HIP_LIS_GPUWIDE": "0"}, {"SPERR_HIP_LIS_L2": "0"}])
def test_list_kernel_choices_in_a_fresh_process(oracle, knobs):
    """Who decodes which list is a matter of speed, never of bits (round 4; the knobs are read once per process):
    `SPERR_HIP_LIS_HI=0` -- a regular tree through k_lis_mx, where regular trees that k_lis_hi cannot take go
    since k_lis_tables was removed; `SPERR_HIP_LIS_GPUWIDE=0` -- no k_lis_l0 / _l1 / _l2 at all, k_lis_hi decodes every
    list; `SPERR_HIP_LIS_L2=0` -- the 8x8x8 sets' list back with k_lis_hi (round 6 gave it a kernel of its own).  Two chunks of 64 x 64 x 32 at 3 bpp and a stream cut
    short, against the oracle's bits (/root/reference/src/SPECK3D_INT.cpp:99-212)."""
    import subprocess
    import sys
    import tempfile
    shape = (64, 64, 64)
    v = turbulence(shape)
    want = oracle.comp_3d(v, (64, 64, 32), 1, 3.0)
    ref = oracle.decomp_3d(want, True)
    with tempfile.TemporaryDirectory() as td:
        np.save(os.path.join(td, "c.npy"), np.frombuffer(want, dtype=np.uint8))
        np.save(os.path.join(td, "r.npy"), ref)
        code = ("import sys, numpy as np, torch; sys.path.insert(0, %r); from sperr_amd.api import SperrHip; "
                "e = SperrHip(); c = torch.from_numpy(np.load(%r)).cuda(); r = np.load(%r); "
                "d = e.decompress(c, True).cpu().numpy(); "
                "sys.exit(0 if np.array_equal(d.view(np.uint32), r.view(np.uint32)) else 3)"
                % (os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."), os.path.join(td, "c.npy"),
                   os.path.join(td, "r.npy")))
        env = dict(os.environ, **knobs)
        assert subprocess.run([sys.executable, "-c", code], env=env, timeout=300).returncode == 0, knobs


@pytest.mark.parametrize("knobs", [{"SPERR_HIP_MX_WGS": "1"}, {"SPERR_HIP_MX_WGS": "3"}, {"SPERR_HIP_MX_WGS": "4096"}])
def test_mixed_shape_decoders_in_a_fresh_process(oracle, knobs):
    """Chunks whose lists mix set shapes decode through k_lis_mx (round 4: fixed regions of the stream handed out to
    several workgroups per chunk, the walker's state passed from region to region); the knobs are read once per
    process.  `SPERR_HIP_MX_WGS=1`: one workgroup per chunk takes every region itself (no hand-over between
    workgroups at all); `=3`: fewer workgroups than chunks, at least two each; `=4096`: eight per chunk whatever else runs.
    (`SPERR_HIP_LIS_MIXED=0`, the serial walk k_lis_walk, has a test of its own above.)  A volume of
    96 x 75 x 110 in chunks of 64 x 50 x 80 (eight chunks of eight shapes, LIS phases of many regions) at 5 bpp, cut
    short as well, and a slice of 300 x 211 -- against the oracle's bits (/root/reference/src/SPECK3D_INT.cpp:99-326,
    /root/reference/src/SPECK2D_INT.cpp:10-218)."""
    import subprocess
    import sys
    import tempfile
    shape = (110, 75, 96)
    v = turbulence(shape)
    want = oracle.comp_3d(v, (64, 50, 80), 1, 5.0)
    ref = oracle.decomp_3d(want, True)
    img = turbulence((1, 211, 300), dtype=np.float64)[0]
    want2 = oracle.comp_2d(img, 2, 85.0, False)
    cut2 = 26 + (len(want2) - 26) // 2
    ref2 = oracle.decomp_2d(want2, (211, 300), False)
    ref2c = oracle.decomp_2d(want2[:cut2], (211, 300), False)
    with tempfile.TemporaryDirectory() as td:
        np.save(os.path.join(td, "c.npy"), np.frombuffer(want, dtype=np.uint8))
        np.save(os.path.join(td, "r.npy"), ref)
        np.save(os.path.join(td, "c2.npy"), np.frombuffer(want2, dtype=np.uint8))
        np.save(os.path.join(td, "r2.npy"), ref2)
        np.save(os.path.join(td, "r2c.npy"), ref2c)
        code = ("import sys, os, numpy as np, torch; sys.path.insert(0, %r); from sperr_amd.api import SperrHip; td = %r; "
                "L = lambda n: np.load(os.path.join(td, n)); e = SperrHip(); "
                "d = e.decompress(torch.from_numpy(L('c.npy')).cuda(), True).cpu().numpy(); "
                "ok = np.array_equal(d.view(np.uint32), L('r.npy').view(np.uint32)); "
                "c2 = L('c2.npy'); "
                "d2 = e.decompress_2d(torch.from_numpy(c2).cuda(), (211, 300), False).cpu().numpy(); "
                "ok = ok and np.array_equal(d2.view(np.uint64), L('r2.npy').view(np.uint64)); "
                "d3 = e.decompress_2d(torch.from_numpy(c2[:%d].copy()).cuda(), (211, 300), False).cpu().numpy(); "
                "ok = ok and np.array_equal(d3.view(np.uint64), L('r2c.npy').view(np.uint64)); "
                "sys.exit(0 if ok else 3)"
                % (os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."), td, cut2))
        env = dict(os.environ, **knobs)
        assert subprocess.run([sys.executable, "-c", code], env=env, timeout=300).returncode == 0, knobs


@pytest.mark.parametrize("shape", [(8, 8), (5, 300), (2, 9), (1, 40), (33, 17), (500, 301), (1024, 1024)])
def test_2d_slice_decoder_on_the_shared_forest(eng, oracle, shape):
    """Slices are decoded by the 3D decoder's kernels on the 2D coder's forest (quadtrees whose
    children come from the bottom right backwards, the subbands released by the type-I set at the end of
    a sorting pass: src/SPECK2D_INT.cpp:44-186): slices without a transform level (no type-I set), with
    empty subbands, one row only, more than 32 bit planes, and streams cut short."""
    img = turbulence((1,) + shape, dtype=np.float64)[0]
    for mode, quality in [(1, 3.0), (2, 120.0), (3, 1e-11)]:
        want = oracle.comp_2d(img, mode, quality, False)
        for cut in (len(want), 26 + (len(want) - 26) // 3):
            s = want[:cut]
            got = eng.decompress_2d(cuda(np.frombuffer(s, dtype=np.uint8)), shape, False).cpu().numpy()
            assert np.array_equal(bits(got), bits(oracle.decomp_2d(s, shape, False))), (shape, mode, cut)


def test_2d_quadtree_walk_decoder_in_a_fresh_process(oracle):
    """`SPERR_HIP_SLICE_MIXED=0` (read once per process) codes slices with k_speck2d's quadtree walk, which
    also decodes the slices the shared forest does not take: the same stream, the same values."""
    import subprocess
    import sys
    import tempfile
    shape = (121, 96)
    img = turbulence((1,) + shape)[0]
    want = oracle.comp_2d(img, 2, 90.0, False)
    ref = oracle.decomp_2d(want, shape, True)
    with tempfile.TemporaryDirectory() as td:
        np.save(os.path.join(td, "c.npy"), np.frombuffer(want, dtype=np.uint8))
        np.save(os.path.join(td, "r.npy"), ref)
        np.save(os.path.join(td, "i.npy"), img)
        code = ("import sys, numpy as np, torch; sys.path.insert(0, %r); from sperr_amd.api import SperrHip; "
                "e = SperrHip(); c = torch.from_numpy(np.load(%r)).cuda(); r = np.load(%r); "
                "d = e.decompress_2d(c, %r, True).cpu().numpy(); "
                "s = e.compress_2d(torch.from_numpy(np.load(%r)).cuda(), 90.0, mode=2, header=False); "
                "ok = np.array_equal(d.view(np.uint32), r.view(np.uint32)) and torch.equal(s, c); "
                "sys.exit(0 if ok else 3)"
                % (os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."), os.path.join(td, "c.npy"),
                   os.path.join(td, "r.npy"), shape, os.path.join(td, "i.npy")))
        env = dict(os.environ, SPERR_HIP_SLICE_MIXED="0")
        assert subprocess.run([sys.executable, "-c", code], env=env, timeout=300).returncode == 0


@pytest.mark.parametrize("shape,chunks", [((64, 64, 64), (16, 16, 16)), ((48, 80, 160), (16, 16, 16)),
                                          ((70, 96, 96), (16, 24, 16))])
@pytest.mark.parametrize("bpp", [0.8, 4.0, 20.0])
def test_many_chunks_of_one_shape_coded_in_two_parts(eng, oracle, shape, chunks, bpp):
    """Fixed-rate compression cuts a group of 64 and more equally shaped chunks into two parts that run
    side by side on two streams (compress_impl, `SPERR_HIP_ENC_PARTS`); the chunk streams land in the
    container in chunk order all the same, the 64-bit retry of single chunks (20 bpp) included; 70 / 16
    leaves border chunks of a second shape beside the 144 regular ones."""
    v = turbulence(shape)
    want = oracle.comp_3d(v, chunks, 1, bpp)
    got = bytes(eng.compress(cuda(v), chunks, bpp).cpu().numpy())
    assert got == want
    dev = cuda(np.frombuffer(want, dtype=np.uint8))
    assert np.array_equal(bits(eng.decompress(dev, True).cpu().numpy()), bits(oracle.decomp_3d(want, True)))


@pytest.mark.parametrize("noisy", ["x_border", "interior"])
def test_more_side_by_side_groups_than_sub_streams(eng, oracle, noisy):
    """88^3 in 16^3 chunks: 64 regular chunks in four parts + seven border shapes = ELEVEN groups side by side
    on eight sub-streams, so groups 0/8, 1/9 and 2/10 share a stream.  Every group reads its own plane
    bounds back (a pinned word pair and an event per group, compress_impl): the field is smooth in one of the
    two groups of such a pair and white noise in the other, so their lowest reachable planes differ by a
    dozen -- a group that took its stream mate's bounds would stop coding early (or start below its top
    plane) and the container would differ from the reference's (src/SPECK_INT.cpp:146-158)."""
    shape = (88, 88, 88)
    v = smooth_field(shape, seed=11, passes=4).astype(np.float32)
    rng = np.random.default_rng(5)
    noise = (rng.standard_normal(shape) * float(np.abs(v).max())).astype(np.float32)
    if noisy == "x_border":      # the border groups 7..10 (x extent 24) noisy, the regular parts 0..3 smooth
        v[:, :, 64:] = noise[:, :, 64:]
    else:                        # the other way round
        v[:64, :64, :64] = noise[:64, :64, :64]
    for bpp in (1.0, 3.0):
        want = oracle.comp_3d(v, (16, 16, 16), 1, bpp)
        got = bytes(eng.compress(cuda(v), (16, 16, 16), bpp).cpu().numpy())
        assert got == want
    dev = cuda(np.frombuffer(want, dtype=np.uint8))
    assert np.array_equal(bits(eng.decompress(dev, True).cpu().numpy()), bits(oracle.decomp_3d(want, True)))


def test_2d_host_entry_points_from_several_threads(eng, oracle):
    """sperr_comp_2d / sperr_decomp_2d called from four host threads at once (each call takes an engine of
    the device's pool and works on a stream and device buffers of the calling thread): every thread gets
    the oracle's stream and values, call after call."""
    import ctypes as C
    import threading
    lib = eng.lib
    lib.sperr_comp_2d.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.c_int, C.c_double, C.c_int,
                                  C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    lib.sperr_decomp_2d.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_size_t, C.c_size_t, C.POINTER(C.c_void_p)]
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    shapes = [(150, 211), (96, 96), (33, 300), (257, 129)]
    imgs = [turbulence((1,) + sh, seed=7 + i)[0] for i, sh in enumerate(shapes)]
    wants = [oracle.comp_2d(im, 2, 80.0, False) for im in imgs]
    decs = [oracle.decomp_2d(w, sh, False) for w, sh in zip(wants, shapes)]
    bad = []

    def work(i):
        dy, dx = shapes[i]
        for _ in range(3):
            dst, ln = C.c_void_p(None), C.c_size_t(0)
            rc = lib.sperr_comp_2d(imgs[i].ctypes.data, 1, dx, dy, 2, 80.0, 0, C.byref(dst), C.byref(ln))
            if rc != 0 or C.string_at(dst.value, ln.value) != wants[i]:
                bad.append(("compress", i, rc))
            libc.free(dst)
            out = C.c_void_p(None)
            rc = lib.sperr_decomp_2d(wants[i], len(wants[i]), 0, dx, dy, C.byref(out))
            got = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_double)), shape=(dy, dx)).copy() if rc == 0 else None
            if rc != 0 or not np.array_equal(bits(got), bits(decs[i])):
                bad.append(("decompress", i, rc))
            if out.value:
                libc.free(out)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(len(shapes))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not bad, bad


@pytest.mark.parametrize("shape,chunks", [((64, 64, 64), (16, 16, 16)), ((48, 80, 160), (16, 16, 16))])
@pytest.mark.parametrize("tol", [3e-2, 1e-4])
def test_many_chunks_pwe_in_two_sub_batches(eng, oracle, shape, chunks, tol):
    """Point-wise error mode with 64 and more chunks of one shape: the decoder runs two sub-batches, each
    with its own 1D decoder of the outlier streams beside it (per sub-batch buffers and events)."""
    v = turbulence(shape)
    want = oracle.comp_3d(v, chunks, 3, tol)
    assert bytes(eng.compress(cuda(v), chunks, tol, mode=3).cpu().numpy()) == want
    dev = cuda(np.frombuffer(want, dtype=np.uint8))
    for as_float in (True, False):
        assert np.array_equal(bits(eng.decompress(dev, as_float).cpu().numpy()),
                              bits(oracle.decomp_3d(want, as_float)))


@pytest.mark.parametrize("shape,chunks", [((64, 64, 64), (16, 16, 16)), ((48, 80, 160), (16, 16, 16))])
@pytest.mark.parametrize("psnr", [60.0, 110.0])
def test_many_chunks_psnr(eng, oracle, shape, chunks, psnr):
    """PSNR mode with 64 and more chunks of one shape (one batch on the encoder's side, two sub-batches
    on the decoder's)."""
    v = turbulence(shape)
    want = oracle.comp_3d(v, chunks, 2, psnr)
    for _ in range(2):
        assert bytes(eng.compress(cuda(v), chunks, psnr, mode=2).cpu().numpy()) == want
    dev = cuda(np.frombuffer(want, dtype=np.uint8))
    assert np.array_equal(bits(eng.decompress(dev, False).cpu().numpy()), bits(oracle.decomp_3d(want, False)))


def test_build_then_smoke_in_one_process():
    """`python __graft_entry__.py smoke` builds, loads the library and then runs the smoke check in the same
    process (the library must not bring a second HIP runtime in before torch has loaded its own)."""
    import subprocess
    import sys
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    r = subprocess.run([sys.executable, os.path.join(root, "__graft_entry__.py"), "smoke"], capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "smoke ok" in r.stdout

"""CPU: the plain-C oracle restatement against the committed golden vectors (generated from the
real reference by tests/golden/make_golden.py) and against the known answers that the reference's
own unit tests pin for the geometry helpers (test_scripts/sperr_helper_unit_test.cpp:7-18,
104-121,219-251; test_scripts/speck3d_flt_unit_test.cpp:29 -- constant field = 17 bytes)."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

from fields import smooth_field

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
with open(os.path.join(GOLD, "golden.json")) as f:
    _G = json.load(f)
    CASES, CASES_2D = _G["cases"], _G["cases_2d"]


def load_input(case):
    name, shape = case["input"], tuple(case["shape_zyx"])
    path = os.path.join(GOLD, name + ".f32")
    if os.path.exists(path):
        arr = np.fromfile(path, dtype=np.float32).reshape(shape)
    else:
        arr = smooth_field(shape, dtype=np.dtype(case["dtype"]))
    assert hashlib.sha256(arr.tobytes()).hexdigest() == case["input_sha256"]
    return arr


@pytest.mark.parametrize("case", CASES, ids=[c["tag"] for c in CASES])
def test_oracle_matches_golden_stream_and_decode(oracle, case):
    arr = load_input(case)
    with open(os.path.join(GOLD, case["tag"] + ".sperr"), "rb") as f:
        want = f.read()
    got = oracle.comp_3d(arr, case["chunks_xyz"], case.get("mode", 1), case["bpp"])
    assert len(got) == case["stream_len"]
    assert got == want  # bit-exact container
    dec_f = oracle.decomp_3d(want, True)
    dec_d = oracle.decomp_3d(want, False)
    assert hashlib.sha256(dec_f.tobytes()).hexdigest() == case["decoded_f32_sha256"]
    assert hashlib.sha256(dec_d.tobytes()).hexdigest() == case["decoded_f64_sha256"]


def load_slice(case):
    name, shape = case["input"], tuple(case["shape_yx"])
    path = os.path.join(GOLD, name + ".f32")
    if os.path.exists(path):
        arr = np.fromfile(path, dtype=np.float32).reshape(shape)
    else:
        arr = smooth_field((1,) + shape, dtype=np.dtype(case["dtype"]))[0]
    assert hashlib.sha256(arr.tobytes()).hexdigest() == case["input_sha256"]
    return arr


@pytest.mark.parametrize("case", CASES_2D, ids=[c["tag"] for c in CASES_2D])
def test_oracle_matches_golden_2d(oracle, case):
    """sperr_comp_2d / sperr_decomp_2d (src/SPERR_C_API.cpp:7-134) through SPECK2D_INT."""
    arr = load_slice(case)
    with open(os.path.join(GOLD, case["tag"] + ".sperr"), "rb") as f:
        want = f.read()
    got = oracle.comp_2d(arr, case["mode"], case["quality"], case["header"])
    assert len(got) == case["stream_len"]
    assert got == want
    body = want[10:] if case["header"] else want
    dec_f = oracle.decomp_2d(body, arr.shape, True)
    dec_d = oracle.decomp_2d(body, arr.shape, False)
    assert hashlib.sha256(dec_f.tobytes()).hexdigest() == case["decoded_f32_sha256"]
    assert hashlib.sha256(dec_d.tobytes()).hexdigest() == case["decoded_f64_sha256"]


def test_dyadic_known_answers(oracle):
    L = oracle.lib
    lev = C.c_size_t(0)
    def dy(d):
        ok = L.orc_can_use_dyadic((C.c_size_t * 3)(*d), C.byref(lev))
        return lev.value if ok else None
    assert dy((64, 1, 1)) is None and dy((64, 64, 1)) is None
    assert dy((64, 64, 64)) == 3 and dy((128, 128, 128)) == 4 and dy((256, 256, 256)) == 5
    assert dy((288, 288, 288)) == 6
    assert dy((256, 256, 300)) == 5 and dy((300, 300, 256)) == 5 and dy((256, 300, 256)) == 5


def test_approx_detail_known_answers(oracle):
    a, d = C.c_size_t(0), C.c_size_t(0)
    for (n, lev), want in {(7, 0): (7, 0), (7, 1): (4, 3), (8, 1): (4, 4), (8, 2): (2, 2),
                           (16, 2): (4, 4)}.items():
        oracle.lib.orc_approx_detail_len(n, lev, C.byref(a), C.byref(d))
        assert (a.value, d.value) == want


def test_chunk_volume_known_answers(oracle):
    ch = oracle.chunk_volume((4, 4, 4), (1, 2, 3))
    assert ch.shape == (8, 6)
    assert ch[0].tolist() == [0, 1, 0, 2, 0, 4] and ch[3].tolist() == [3, 1, 0, 2, 0, 4]
    assert ch[4].tolist() == [0, 1, 2, 2, 0, 4] and ch[7].tolist() == [3, 1, 2, 2, 0, 4]
    ch = oracle.chunk_volume((4, 4, 1), (1, 2, 3))
    assert ch.shape == (8, 6) and ch[7].tolist() == [3, 1, 2, 2, 0, 1]


def test_constant_chunk_is_17_bytes(oracle):
    v = np.full((12, 10, 9), 3.25, dtype=np.float64)
    s = oracle.chunk_compress_rate(v, 2.0)
    assert len(s) == 17 and s[0] == 0x81
    assert np.array_equal(oracle.chunk_decompress(s, v.shape), v)


def test_speck_lossless_roundtrip(oracle):
    """Mirror of test_scripts/speck_int_unit_test.cpp:503-713: an unbudgeted encode/decode returns
    the coefficients and signs exactly."""
    rng = np.random.default_rng(1)
    for shape in [(4, 4, 4), (7, 9, 13), (16, 16, 16), (5, 33, 20)]:
        coef = np.rint(np.abs(rng.normal(0, 300, size=shape))).astype(np.uint64)
        coef[rng.random(shape) < 0.3] = 0
        n = coef.size
        signbits = rng.integers(0, 2, size=n).astype(np.uint64)
        sign = np.zeros((n + 63) // 64, dtype=np.uint64)
        for i in np.nonzero(signbits)[0]:
            sign[i >> 6] |= np.uint64(1) << np.uint64(i & 63)
        s = oracle.speck3d_encode(coef, sign, 0)
        c2, s2 = oracle.speck3d_decode(s, shape)
        assert np.array_equal(c2, coef)
        nz = coef.reshape(-1) != 0
        got = np.array([(int(s2[i >> 6]) >> (i & 63)) & 1 for i in range(n)], dtype=np.uint64)
        assert np.array_equal(got[nz], signbits[nz])


def test_dwt_roundtrip(oracle):
    """After test_scripts/dwt_unit_test.cpp:262-362: DWT then IDWT gives the input back (to fp64
    round-off) for even and odd sizes, dyadic and wavelet-packet."""
    for shape in [(17, 17, 17), (16, 20, 32), (23, 45, 70), (9, 64, 64)]:
        a = smooth_field(shape)
        back = oracle.idwt3d(oracle.dwt3d(a.astype(np.float64)))
        assert np.max(np.abs(back - a)) <= 1e-11 * np.max(np.abs(a))

"""CPU: the data-parallel SPECK3D formulation (tests/model/speck_model.cpp, which shares
sperr_amd/csrc/speck_tree.h with the HIP kernels) against the oracle, bit for bit."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from sperr_amd.synth import turbulence

HERE = os.path.dirname(os.path.abspath(__file__))
_sz, _vp = C.c_size_t, C.c_void_p


@pytest.fixture(scope="module")
def model():
    so = os.path.join(HERE, "model", "libspeck_model.so")
    src = os.path.join(HERE, "model", "speck_model.cpp")
    deps = [src, os.path.join(HERE, "..", "sperr_amd", "csrc", "speck_tree.h"),
            os.path.join(HERE, "..", "sperr_amd", "csrc", "speck_tree_host.hpp"),
            os.path.join(HERE, "..", "sperr_amd", "csrc", "bit_words.h")]
    if not os.path.exists(so) or any(os.path.getmtime(so) < os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so, src])
    lib = C.CDLL(so)
    lib.model_speck3d_encode.argtypes = [_vp, _vp, _vp, _sz, C.POINTER(_vp), C.POINTER(_sz)]
    lib.model_speck3d_encode.restype = C.c_int
    if hasattr(lib, "model_speck3d_decode"):
        lib.model_speck3d_decode.argtypes = [_vp, _sz, _vp, _vp, _vp]
        lib.model_speck3d_decode.restype = C.c_int
    return lib


def model_encode(lib, coef, sign, budget):
    c = np.ascontiguousarray(coef, dtype=np.uint64)
    dz, dy, dx = c.shape
    out, n = _vp(None), _sz(0)
    lib.model_speck3d_encode(c.ctypes.data, sign.ctypes.data, (_sz * 3)(dx, dy, dz), budget,
                             C.byref(out), C.byref(n))
    s = C.string_at(out.value, n.value)
    C.CDLL(None).free(out)
    return s


def quantized(oracle, shape, scale):
    v = oracle.dwt3d(turbulence(shape).astype(np.float64))
    q = np.abs(v).max() / scale
    coef, sign, _ = oracle.quantize(v, q)
    return coef, sign


SHAPES = [(8, 8, 8), (16, 16, 16), (17, 17, 17), (13, 21, 30), (32, 32, 32), (9, 40, 48),
          (41, 64, 64), (3, 5, 7), (1, 16, 16), (2, 2, 2), (48, 48, 48)]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("budget", [0, 1000, 20000])
def test_model_encoder_matches_oracle(oracle, model, shape, budget):
    coef, sign = quantized(oracle, shape, 3000.0)
    want = oracle.speck3d_encode(coef, sign, budget)
    got = model_encode(model, coef, sign, budget)
    assert got[:9] == want[:9]
    assert got == want


def test_model_encoder_sparse_and_zero(oracle, model):
    coef = np.zeros((12, 12, 12), dtype=np.uint64)
    sign = np.full((coef.size + 63) // 64, np.uint64(0xFFFFFFFFFFFFFFFF))
    assert model_encode(model, coef, sign, 0) == oracle.speck3d_encode(coef, sign, 0)
    coef[3, 4, 5] = 77
    coef[11, 0, 2] = 1
    assert model_encode(model, coef, sign, 0) == oracle.speck3d_encode(coef, sign, 0)
    coef[0, 0, 0] = (1 << 52) + 12345
    assert model_encode(model, coef, sign, 0) == oracle.speck3d_encode(coef, sign, 0)


def model_decode(lib, stream, shape):
    dz, dy, dx = shape
    buf = np.frombuffer(stream, dtype=np.uint8)
    coef = np.zeros(shape, dtype=np.uint64)
    sign = np.zeros((coef.size + 63) // 64, dtype=np.uint64)
    lib.model_speck3d_decode(buf.ctypes.data, buf.size, (_sz * 3)(dx, dy, dz), coef.ctypes.data,
                             sign.ctypes.data)
    return coef, sign


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("budget", [0, 1000, 20000])
def test_model_decoder_matches_oracle(oracle, model, shape, budget):
    coef, sign = quantized(oracle, shape, 3000.0)
    stream = oracle.speck3d_encode(coef, sign, budget)
    for cut in (len(stream), 9 + (len(stream) - 9) // 2, 9 + (len(stream) - 9) // 7):
        s = stream[:cut]
        c0, s0 = oracle.speck3d_decode(s, shape)
        c1, s1 = model_decode(model, s, shape)
        assert np.array_equal(c0, c1)
        assert np.array_equal(s0, s1)


def model_decode_par(lib, stream, shape, window):
    dz, dy, dx = shape
    buf = np.frombuffer(stream, dtype=np.uint8)
    coef = np.zeros(shape, dtype=np.uint64)
    sign = np.zeros((coef.size + 63) // 64, dtype=np.uint64)
    lib.model_speck3d_decode_par.argtypes = [_vp, _sz, _vp, _vp, _vp, C.c_int]
    lib.model_speck3d_decode_par(buf.ctypes.data, buf.size, (_sz * 3)(dx, dy, dz),
                                 coef.ctypes.data, sign.ctypes.data, window)
    return coef, sign


@pytest.mark.parametrize("shape", [(8, 8, 8), (16, 16, 16), (32, 32, 32), (64, 64, 64),
                                   (16, 16, 32), (32, 16, 16), (13, 21, 30), (17, 17, 17)])
@pytest.mark.parametrize("window", [64, 200, 4096])
def test_model_table_driven_decoder_matches_oracle(oracle, model, shape, window):
    """The speculative per-position table + hop + parallel-expand formulation of the LIS phase
    (regular levels), with tiny windows to stress every boundary case."""
    for scale in (3000.0, 4294967295.0):
        coef, sign = quantized(oracle, shape, scale)
        for budget in (0, 30000):
            stream = oracle.speck3d_encode(coef, sign, budget)
            for cut in (len(stream), 9 + (len(stream) - 9) // 3):
                s = stream[:cut]
                c0, s0 = oracle.speck3d_decode(s, shape)
                c1, s1 = model_decode_par(model, s, shape, window)
                assert np.array_equal(c0, c1)
                assert np.array_equal(s0, s1)


def model_decode_mixed(lib, stream, shape, window, hmax):
    dz, dy, dx = shape
    buf = np.frombuffer(stream, dtype=np.uint8)
    coef = np.zeros(shape, dtype=np.uint64)
    sign = np.zeros((coef.size + 63) // 64, dtype=np.uint64)
    lib.model_speck3d_decode_mixed.argtypes = [_vp, _sz, _vp, _vp, _vp, C.c_int, C.c_int]
    lib.model_speck3d_decode_mixed.restype = C.c_int
    rc = lib.model_speck3d_decode_mixed(buf.ctypes.data, buf.size, (_sz * 3)(dx, dy, dz),
                                        coef.ctypes.data, sign.ctypes.data, window, hmax)
    assert rc == 0
    return coef, sign


MIXED_SHAPES = [(17, 17, 17), (13, 21, 30), (3, 5, 7), (1, 16, 16), (2, 2, 2), (48, 48, 48), (1, 1, 1),
                (9, 40, 48), (41, 64, 64), (25, 25, 25), (31, 33, 30), (20, 18, 16), (12, 12, 12),
                (1, 1, 9), (5, 1, 1), (16, 16, 16), (50, 35, 17)]


@pytest.mark.parametrize("shape", MIXED_SHAPES)
@pytest.mark.parametrize("window,hmax", [(64, 0), (150, 1), (4096, 1), (1200, 2), (300, -1)])
def test_model_mixed_shape_decoder_matches_oracle(oracle, model, shape, window, hmax):
    """Shape-class tables + one serial walk + parallel expansion (the formulation of k_lis_mx and of its predecessor k_lis_mixed) on
    chunks whose lists mix set shapes: odd lengths, wavelet-packet shapes, tiny chunks; small
    windows stress the window boundaries, hmax = -1 walks into every set (no tables at all)."""
    for scale in (3000.0, 4294967295.0):
        coef, sign = quantized(oracle, shape, scale)
        for budget in (0, 30000):
            stream = oracle.speck3d_encode(coef, sign, budget)
            for cut in (len(stream), 9 + (len(stream) - 9) // 3):
                s = stream[:cut]
                c0, s0 = oracle.speck3d_decode(s, shape)
                c1, s1 = model_decode_mixed(model, s, shape, window, hmax)
                assert np.array_equal(c0, c1)
                assert np.array_equal(s0, s1)


def test_shape_classes_are_consistent(model):
    """Host-side shape classes (sperr_amd/csrc/speck_tree_host.hpp build_classes): for every set of the
    forest the class read off the node (spk::node_cls) has as many children as the node, the children's
    classes are those of the child nodes, leaf parents sit in columns 1..3 by their size and every class
    with a column has columns for all its children."""
    lib = model
    lib.model_check_classes.argtypes = [_vp]
    lib.model_check_classes.restype = C.c_int
    for dims in [(250, 250, 250), (129, 129, 129), (100, 70, 33), (96, 96, 96), (17, 300, 21), (1, 1, 9),
                 (80, 80, 80), (64, 64, 320), (30, 40, 8), (2, 3, 5)]:
        assert lib.model_check_classes((_sz * 3)(*dims)) == 0, dims
    lib.model_check_classes_2d.argtypes = [_vp]
    lib.model_check_classes_2d.restype = C.c_int
    for dims in [(999, 999, 1), (64, 64, 1), (17, 23, 1), (300, 9, 1), (128, 96, 1), (1, 50, 1), (1024, 1000, 1)]:
        assert lib.model_check_classes_2d((_sz * 3)(*dims)) == 0, dims   # (children in the 2D coder's order)


def test_mx_columns(model):
    """The columns of k_lis_mx (spk::build_mx_columns): one class per column, children in lower columns, numbered by
    steps above the leaf parents, two different column groups per list level; a chunk with three ragged axes (232 =
    40 x 4 + 24 x 3 per axis: eight classes of 4x4x4-sized sets of comparable frequency) gets all twelve, and so do a
    250^3 chunk and a 999 x 999 slice."""
    lib = model
    lib.model_check_mx_columns.argtypes = [_vp, C.c_int, C.POINTER(C.c_int)]
    lib.model_check_mx_columns.restype = C.c_int
    n = C.c_int(0)
    for dims in [(250, 250, 250), (232, 232, 232), (232, 256, 256), (129, 129, 129), (100, 70, 33), (17, 300, 21),
                 (64, 64, 64), (30, 40, 8), (2, 3, 5), (1, 1, 9)]:
        assert lib.model_check_mx_columns((_sz * 3)(*dims), 0, C.byref(n)) == 0, dims
        if dims in ((250, 250, 250), (232, 232, 232)):
            assert n.value == 12, (dims, n.value)
        elif min(dims) >= 100:   # (one ragged axis: two classes per step)
            assert n.value >= 6, (dims, n.value)
    for dims in [(999, 999, 1), (64, 64, 1), (17, 23, 1), (300, 9, 1), (1, 50, 1)]:
        assert lib.model_check_mx_columns((_sz * 3)(*dims), 1, C.byref(n)) == 0, dims
    assert lib.model_check_mx_columns((_sz * 3)(999, 999, 1), 1, C.byref(n)) == 0 and n.value == 12


def _speck2d_oracle(oracle, coef2d, sign, budget):
    lib = oracle.lib
    lib.orc_speck2d_encode.argtypes = [_vp, _vp, _sz, _sz, _sz, C.POINTER(_vp), C.POINTER(_sz)]
    lib.orc_speck2d_encode.restype = C.c_int
    lib.orc_speck2d_decode.argtypes = [_vp, _sz, _sz, _sz, _vp, _vp]
    lib.orc_speck2d_decode.restype = C.c_int
    dy, dx = coef2d.shape
    out, n = _vp(None), _sz(0)
    c = np.ascontiguousarray(coef2d, dtype=np.uint64)
    assert lib.orc_speck2d_encode(c.ctypes.data, sign.ctypes.data, dx, dy, budget, C.byref(out), C.byref(n)) == 0
    s = C.string_at(out.value, n.value)
    C.CDLL(None).free(out)
    return s


def _speck2d_oracle_decode(oracle, stream, shape_yx):
    dy, dx = shape_yx
    buf = np.frombuffer(stream, dtype=np.uint8)
    coef = np.zeros(shape_yx, dtype=np.uint64)
    sign = np.zeros((coef.size + 63) // 64, dtype=np.uint64)
    assert oracle.lib.orc_speck2d_decode(buf.ctypes.data, buf.size, dx, dy, coef.ctypes.data, sign.ctypes.data) == 0
    return coef, sign


@pytest.mark.parametrize("shape", [(16, 16), (17, 23), (64, 64), (99, 100), (40, 9), (128, 96), (9, 9), (200, 33)])
@pytest.mark.parametrize("window,hmax", [(200, 1), (4096, 2), (64, 0)])
def test_model_2d_decoder_on_the_mixed_machinery(oracle, model, shape, window, hmax):
    """The 2D coder's streams (SPECK2D_INT: quadtrees, children from the bottom right backwards, the
    subbands released by the type-I set at the end of a pass) decoded with the forest / shape-class /
    window machinery of the 3D decoder in its 2D mode, against the oracle's SPECK2D decoder."""
    lib = model
    lib.model_speck2d_decode_mixed.argtypes = [_vp, _sz, _vp, _vp, _vp, C.c_int, C.c_int]
    lib.model_speck2d_decode_mixed.restype = C.c_int
    dy, dx = shape
    for scale in (3000.0, 4294967295.0):
        coef3, sign = quantized(oracle, (1, dy, dx), scale)
        coef = coef3.reshape(dy, dx)
        for budget in (0, 20000):
            stream = _speck2d_oracle(oracle, coef, sign, budget)
            for cut in (len(stream), 9 + (len(stream) - 9) // 3):
                s = stream[:cut]
                c0, s0 = _speck2d_oracle_decode(oracle, s, shape)
                buf = np.frombuffer(s, dtype=np.uint8)
                c1 = np.zeros(shape, dtype=np.uint64)
                s1 = np.zeros((c1.size + 63) // 64, dtype=np.uint64)
                assert lib.model_speck2d_decode_mixed(buf.ctypes.data, buf.size, (_sz * 3)(dx, dy, 1),
                                                      c1.ctypes.data, s1.ctypes.data, window, hmax) == 0
                assert np.array_equal(c0, c1), (shape, scale, budget, cut)
                assert np.array_equal(s0, s1)


@pytest.mark.parametrize("shape", [(16, 16), (17, 23), (64, 64), (99, 100), (40, 9), (128, 96), (9, 9), (200, 33), (8, 8), (5, 300)])
@pytest.mark.parametrize("budget", [0, 1500, 40000])
def test_model_2d_encoder_on_the_shared_forest(oracle, model, shape, budget):
    """The count -> scan -> scatter formulation of the 3D encoder on the 2D coder's forest, with the
    type-I phase the kernel k_enc_iphase runs (tests/model/speck_model.cpp::model_speck2d_encode):
    the oracle's SPECK2D stream, byte for byte."""
    lib = model
    lib.model_speck2d_encode.argtypes = [_vp, _vp, _vp, _sz, C.POINTER(_vp), C.POINTER(_sz)]
    lib.model_speck2d_encode.restype = C.c_int
    dy, dx = shape
    for scale in (3000.0, 4294967295.0):
        coef3, sign = quantized(oracle, (1, dy, dx), scale)
        coef = np.ascontiguousarray(coef3.reshape(dy, dx), dtype=np.uint64)
        want = _speck2d_oracle(oracle, coef, sign, budget)
        out, n = _vp(None), _sz(0)
        assert lib.model_speck2d_encode(coef.ctypes.data, sign.ctypes.data, (_sz * 3)(dx, dy, 1), budget,
                                        C.byref(out), C.byref(n)) == 0
        got = C.string_at(out.value, n.value)
        C.CDLL(None).free(out)
        assert got[:9] == want[:9], (shape, scale, budget)
        assert got == want, (shape, scale, budget)


@pytest.mark.parametrize("n,k,top", [(4, 2, 3), (5, 5, 9), (100, 7, 200), (1000, 1000, 5), (4097, 300, 5),
                                     (65536, 1000, 70000), (50000, 20000, 3), (1 << 18, 3000, 2), (777, 1, 1)])
def test_model_1d_coder_paths(oracle, model, n, k, top):
    """The coder of the outlier list in the kernels' formulations (tests/model/speck_model.cpp::
    model_speck1d_encode / _decode): the encoder gives every outlier of a significant run one path found from
    its position and its neighbours, the decoder parses a whole path per step with the run lengths in closed
    form -- the oracle's stream byte for byte, and the values back from it.  The batched decoder model also works R out of
    the raw records of runs of 2^g values with the scan k_speck1d's flush_paths uses for the asm chain's records
    (sperr_amd/csrc/outlier.hip, chain_p2) and checks it against the R the chain carried along (-7 / -8 if not)."""
    from oracle.pyoracle import pack_mask, unpack_mask
    lib = model
    lib.model_speck1d_encode.argtypes = [_vp, _vp, _sz, C.POINTER(_vp), C.POINTER(_sz)]
    lib.model_speck1d_encode.restype = C.c_int
    lib.model_speck1d_decode.argtypes = [_vp, _sz, _sz, _vp, _vp]
    lib.model_speck1d_decode.restype = C.c_int
    rng = np.random.default_rng(n + k)
    for rep in range(3):
        coef = np.zeros(n, dtype=np.uint64)
        pos = rng.choice(n, size=min(k, n), replace=False)
        if rep == 2:   # clustered outliers
            pos = np.unique(np.clip(pos // 7 + n // 3, 0, n - 1))
        coef[pos] = rng.integers(1, top + 1, size=pos.size, dtype=np.uint64)
        sign = rng.integers(0, 2, size=n).astype(bool)
        want = oracle.speck1d_encode(coef, sign)
        sm = pack_mask(sign)
        out, ln = _vp(None), _sz(0)
        assert lib.model_speck1d_encode(coef.ctypes.data, sm.ctypes.data, n, C.byref(out), C.byref(ln)) == 0
        got = C.string_at(out.value, ln.value)
        C.CDLL(None).free(out)
        assert got == want, (n, k, top, rep)
        buf = np.frombuffer(want, dtype=np.uint8)
        c2 = np.zeros(n, dtype=np.uint64)
        s2 = np.zeros((n + 63) // 64, dtype=np.uint64)
        assert lib.model_speck1d_decode(buf.ctypes.data, buf.size, n, c2.ctypes.data, s2.ctypes.data) == 0
        assert np.array_equal(c2, coef), (n, k, top, rep)
        assert np.array_equal(unpack_mask(s2, n)[coef > 0], sign[coef > 0])
        # the chain + batched-records formulation (k_speck1d<false> since round 3)
        lib.model_speck1d_decode_batched.argtypes = [_vp, _sz, _sz, _vp, _vp, C.c_uint32]
        lib.model_speck1d_decode_batched.restype = C.c_int
        for batch in (64, 3):
            c3 = np.zeros(n, dtype=np.uint64)
            s3 = np.zeros((n + 63) // 64, dtype=np.uint64)
            assert lib.model_speck1d_decode_batched(buf.ctypes.data, buf.size, n, c3.ctypes.data, s3.ctypes.data,
                                                    batch) == 0
            assert np.array_equal(c3, coef), (n, k, top, rep, batch)
            assert np.array_equal(unpack_mask(s3, n)[coef > 0], sign[coef > 0])


def test_model_bit_words(model):
    """The word-level steps of the decoder's pixel passes (sperr_amd/csrc/bit_words.h: bits spread under / gathered
    from under a mask with parallel-suffix steps, a word's LIP token starts with one addition) against bit-by-bit
    loops, on 3 x 200 000 pseudo-random words."""
    model.model_check_bit_words.argtypes = [C.c_uint64, C.c_int]
    model.model_check_bit_words.restype = C.c_int
    for seed in (1, 2, 3):
        assert model.model_check_bit_words(seed, 200000) == 0

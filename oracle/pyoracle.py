"""ctypes bindings for the parity oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
It loads

* ``oracle/_build/libsperr_oracle.so``  -- this repo's plain-C restatement (``Oracle``), and
* ``oracle/_ref/libSPERR_ref.so`` + ``libref_probe.so`` -- the real reference compiled from
  /root/reference by ``oracle/Makefile`` (``Ref``), when those prebuilt files are present.

Nothing here reads /root/reference at run time.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_sz = C.c_size_t
_vp = C.c_void_p


def build_oracle(force=False):
    """Compile the C restatement (gcc) if the .so is missing or stale."""
    so = os.path.join(HERE, "_build", "libsperr_oracle.so")
    src = os.path.join(HERE, "sperr_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", HERE, "all"])
    return so


def build_ref():
    """(Re)build oracle/_ref from /root/reference when that tree is present; otherwise keep the
    prebuilt files.  Returns True when libSPERR_ref.so + libref_probe.so exist afterwards."""
    if os.path.isdir("/root/reference/src"):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])
    return have_ref()


def have_ref():
    return all(os.path.exists(os.path.join(HERE, "_ref", f))
               for f in ("libSPERR_ref.so", "libref_probe.so"))


def pack_mask(bools):
    """bool[n] -> uint64 words, bit i of the mask = bools[i] (the reference's Bitmask layout)."""
    b = np.asarray(bools, dtype=bool).ravel()
    pad = (-b.size) % 64
    if pad:
        b = np.concatenate([b, np.zeros(pad, dtype=bool)])
    return np.packbits(b, bitorder="little").view(np.uint64).copy()


def unpack_mask(words, n):
    return np.unpackbits(np.ascontiguousarray(words).view(np.uint8), bitorder="little")[:n].astype(bool)


def _u8(buf):
    return np.frombuffer(buf, dtype=np.uint8) if not isinstance(buf, np.ndarray) else buf


class _CApiMixin:
    """sperr_comp_3d / sperr_decomp_3d shaped entry points (include/SPERR_C_API.h:106-137)."""

    _comp = None
    _decomp = None
    _trunc = None
    _comp2 = None
    _decomp2 = None
    _decomp2mr = None
    _libc = C.CDLL(None)

    def _wire(self):
        self._comp.restype = C.c_int
        self._comp.argtypes = [_vp, C.c_int, _sz, _sz, _sz, _sz, _sz, _sz, C.c_int, C.c_double, _sz,
                               C.POINTER(_vp), C.POINTER(_sz)]
        self._decomp.restype = C.c_int
        self._decomp.argtypes = [_vp, _sz, C.c_int, _sz, C.POINTER(_sz), C.POINTER(_sz),
                                 C.POINTER(_sz), C.POINTER(_vp)]
        self._libc.free.argtypes = [_vp]
        if self._trunc is not None:
            self._trunc.restype = C.c_int
            self._trunc.argtypes = [_vp, _sz, C.c_uint, C.POINTER(_vp), C.POINTER(_sz)]
        if self._comp2 is not None:
            self._comp2.restype = C.c_int
            self._comp2.argtypes = [_vp, C.c_int, _sz, _sz, C.c_int, C.c_double, C.c_int,
                                    C.POINTER(_vp), C.POINTER(_sz)]
            self._decomp2.restype = C.c_int
            self._decomp2.argtypes = [_vp, _sz, C.c_int, _sz, _sz, C.POINTER(_vp)]

    def trunc_3d(self, stream, pct):
        """sperr_trunc_3d (include/SPERR_C_API.h:151-156). Returns bytes."""
        buf = np.frombuffer(stream, dtype=np.uint8)
        dst, n = _vp(None), _sz(0)
        rtn = self._trunc(buf.ctypes.data, buf.size, pct, C.byref(dst), C.byref(n))
        if rtn != 0:
            raise RuntimeError(f"trunc_3d returned {rtn}")
        out = C.string_at(dst.value, n.value)
        self._libc.free(dst)
        return out

    def comp_2d(self, img, mode, quality, header=False):
        """img: numpy float32/float64 (y, x). Returns bytes (sperr_comp_2d)."""
        img = np.ascontiguousarray(img)
        assert img.dtype in (np.float32, np.float64) and img.ndim == 2
        dy, dx = img.shape
        dst, n = _vp(None), _sz(0)
        rtn = self._comp2(img.ctypes.data, int(img.dtype == np.float32), dx, dy, mode, quality,
                          int(header), C.byref(dst), C.byref(n))
        if rtn != 0:
            raise RuntimeError(f"comp_2d returned {rtn}")
        out = C.string_at(dst.value, n.value)
        self._libc.free(dst)
        return out

    def decomp_2d(self, stream, shape_yx, as_float):
        buf = np.frombuffer(stream, dtype=np.uint8)
        dy, dx = shape_yx
        dst = _vp(None)
        rtn = self._decomp2(buf.ctypes.data, buf.size, int(as_float), dx, dy, C.byref(dst))
        if rtn != 0:
            raise RuntimeError(f"decomp_2d returned {rtn}")
        dt = np.float32 if as_float else np.float64
        out = np.frombuffer(C.string_at(dst.value, dx * dy * np.dtype(dt).itemsize), dtype=dt).reshape(dy, dx).copy()
        self._libc.free(dst)
        return out

    def decomp_2d_multi_res(self, stream, shape_yx):
        """-> (slice float64 (y, x), [level slices float64, coarsest first])."""
        buf = np.frombuffer(stream, dtype=np.uint8)
        dy, dx = shape_yx
        dst, nlev = _vp(None), _sz(0)
        ldims = (_sz * 32)()
        levels = (_vp * 16)()
        self._decomp2mr.restype = C.c_int
        self._decomp2mr.argtypes = [_vp, _sz, _sz, _sz, C.POINTER(_vp), C.POINTER(_sz), _vp, _vp]
        rtn = self._decomp2mr(buf.ctypes.data, buf.size, dx, dy, C.byref(dst), C.byref(nlev), ldims, levels)
        if rtn != 0:
            raise RuntimeError(f"decomp_2d_multi_res returned {rtn}")
        img = np.frombuffer(C.string_at(dst.value, dx * dy * 8), dtype=np.float64).reshape(dy, dx).copy()
        self._libc.free(dst)
        out = []
        for h in range(nlev.value):
            lx, ly = ldims[2 * h], ldims[2 * h + 1]
            out.append(np.frombuffer(C.string_at(levels[h], lx * ly * 8), dtype=np.float64).reshape(ly, lx).copy())
            self._libc.free(_vp(levels[h]))
        return img, out

    def comp_3d(self, vol, chunks, mode, quality, nthreads=1):
        """vol: numpy float32/float64 array shaped (z, y, x). Returns bytes."""
        vol = np.ascontiguousarray(vol)
        assert vol.dtype in (np.float32, np.float64) and vol.ndim == 3
        dz, dy, dx = vol.shape
        dst, n = _vp(None), _sz(0)
        rtn = self._comp(vol.ctypes.data, int(vol.dtype == np.float32), dx, dy, dz, chunks[0],
                         chunks[1], chunks[2], mode, quality, nthreads, C.byref(dst), C.byref(n))
        if rtn != 0:
            raise RuntimeError(f"comp_3d returned {rtn}")
        out = C.string_at(dst.value, n.value)
        self._libc.free(dst)
        return out

    def decomp_3d(self, stream, output_float=True, nthreads=1):
        """Returns a numpy array shaped (z, y, x)."""
        buf = np.frombuffer(stream, dtype=np.uint8)
        dst = _vp(None)
        dx, dy, dz = _sz(0), _sz(0), _sz(0)
        rtn = self._decomp(buf.ctypes.data, buf.size, int(output_float), nthreads, C.byref(dx),
                           C.byref(dy), C.byref(dz), C.byref(dst))
        if rtn != 0:
            raise RuntimeError(f"decomp_3d returned {rtn}")
        n = dx.value * dy.value * dz.value
        dt = np.float32 if output_float else np.float64
        # (ctypes.string_at takes the size as a C int: volumes of 2 GiB and more go through a view)
        view = np.ctypeslib.as_array(C.cast(dst, C.POINTER(C.c_float if output_float else C.c_double)), shape=(n,))
        out = view.astype(dt, copy=True)
        self._libc.free(dst)
        return out.reshape(dz.value, dy.value, dx.value)


class Oracle(_CApiMixin):
    """The plain-C restatement (oracle/sperr_oracle.c)."""

    def __init__(self):
        self.lib = C.CDLL(build_oracle())
        L = self.lib
        self._comp, self._decomp = L.orc_comp_3d, L.orc_decomp_3d
        self._trunc = L.orc_trunc_3d
        self._comp2, self._decomp2 = L.orc_comp_2d, L.orc_decomp_2d
        self._decomp2mr = L.orc_decomp_2d_multi_res
        self._wire()
        L.orc_dwt3d.argtypes = [_vp, _vp]
        L.orc_idwt3d.argtypes = [_vp, _vp]
        L.orc_condition.argtypes = [_vp, _sz, _vp]
        L.orc_condition.restype = C.c_int
        L.orc_inverse_condition.argtypes = [_vp, _sz, _vp]
        L.orc_quantize.argtypes = [_vp, _sz, C.c_double, _vp, _vp, C.POINTER(C.c_int)]
        L.orc_quantize.restype = C.c_int
        L.orc_inv_quantize.argtypes = [_vp, _vp, _sz, C.c_double, _vp]
        L.orc_speck3d_encode.argtypes = [_vp, _vp, _vp, _sz, C.POINTER(_vp), C.POINTER(_sz)]
        L.orc_speck3d_encode.restype = C.c_int
        L.orc_speck3d_decode.argtypes = [_vp, _sz, _vp, _vp, _vp]
        L.orc_speck3d_decode.restype = C.c_int
        L.orc_decomp_3d_multi_res.restype = C.c_int
        L.orc_decomp_3d_multi_res.argtypes = [_vp, _sz, _sz, C.POINTER(_sz), C.POINTER(_sz),
                                              C.POINTER(_sz), C.POINTER(_vp), C.POINTER(_sz), _vp, _vp]
        L.orc_speck1d_encode.argtypes = [_vp, _vp, _sz, C.POINTER(_vp), C.POINTER(_sz)]
        L.orc_speck1d_encode.restype = C.c_int
        L.orc_speck1d_decode.argtypes = [_vp, _sz, _sz, _vp, _vp]
        L.orc_speck1d_decode.restype = C.c_int
        L.orc_chunk_compress_rate.argtypes = [_vp, _vp, C.c_double, C.POINTER(_vp), C.POINTER(_sz)]
        L.orc_chunk_compress_rate.restype = C.c_int
        L.orc_chunk_decompress.argtypes = [_vp, _sz, _vp, _vp]
        L.orc_chunk_decompress.restype = C.c_int
        L.orc_chunk_volume.argtypes = [_vp, _vp, _vp, _sz]
        L.orc_chunk_volume.restype = _sz
        L.orc_num_of_xforms.argtypes = [_sz]
        L.orc_num_of_xforms.restype = _sz
        L.orc_num_of_partitions.argtypes = [_sz]
        L.orc_num_of_partitions.restype = _sz
        L.orc_can_use_dyadic.argtypes = [_vp, C.POINTER(_sz)]
        L.orc_can_use_dyadic.restype = C.c_int
        L.orc_approx_detail_len.argtypes = [_sz, _sz, C.POINTER(_sz), C.POINTER(_sz)]

    @staticmethod
    def _dims(dims):  # dims given as (x, y, z)
        return (_sz * 3)(*dims)

    def dwt3d(self, vol):
        """vol: float64 (z,y,x); returns the transformed copy."""
        v = np.array(vol, dtype=np.float64, order="C")
        dz, dy, dx = v.shape
        self.lib.orc_dwt3d(v.ctypes.data, self._dims((dx, dy, dz)))
        return v

    def idwt3d(self, vol):
        v = np.array(vol, dtype=np.float64, order="C")
        dz, dy, dx = v.shape
        self.lib.orc_idwt3d(v.ctypes.data, self._dims((dx, dy, dz)))
        return v

    def condition(self, vol):
        v = np.array(vol, dtype=np.float64, order="C")
        hdr = np.zeros(17, dtype=np.uint8)
        const = self.lib.orc_condition(v.ctypes.data, v.size, hdr.ctypes.data)
        return v, hdr.tobytes(), bool(const)

    def quantize(self, vals, q):
        v = np.ascontiguousarray(vals, dtype=np.float64)
        coef = np.zeros(v.size, dtype=np.uint64)
        sign = np.zeros((v.size + 63) // 64, dtype=np.uint64)
        w = C.c_int(0)
        rtn = self.lib.orc_quantize(v.ctypes.data, v.size, q, coef.ctypes.data, sign.ctypes.data,
                                    C.byref(w))
        if rtn:
            raise RuntimeError(f"orc_quantize returned {rtn}")
        return coef.reshape(v.shape), sign, w.value

    def inv_quantize(self, coef, sign, q):
        c = np.ascontiguousarray(coef, dtype=np.uint64)
        out = np.zeros(c.shape, dtype=np.float64)
        self.lib.orc_inv_quantize(c.ctypes.data, sign.ctypes.data, c.size, q, out.ctypes.data)
        return out

    def speck3d_encode(self, coef, sign, budget_bits=0):
        """coef: uint64 (z,y,x); sign: uint64 words. Returns the 9-byte-header stream."""
        c = np.ascontiguousarray(coef, dtype=np.uint64)
        dz, dy, dx = c.shape
        out, n = _vp(None), _sz(0)
        self.lib.orc_speck3d_encode(c.ctypes.data, sign.ctypes.data, self._dims((dx, dy, dz)),
                                    budget_bits, C.byref(out), C.byref(n))
        s = C.string_at(out.value, n.value)
        self._libc.free(out)
        return s

    def speck3d_decode(self, stream, shape_zyx):
        dz, dy, dx = shape_zyx
        buf = np.frombuffer(stream, dtype=np.uint8)
        coef = np.zeros(shape_zyx, dtype=np.uint64)
        sign = np.zeros((coef.size + 63) // 64, dtype=np.uint64)
        self.lib.orc_speck3d_decode(buf.ctypes.data, buf.size, self._dims((dx, dy, dz)),
                                    coef.ctypes.data, sign.ctypes.data)
        return coef, sign

    def decomp_3d_multi_res(self, stream, nthreads=1):
        """-> (volume float64 (z, y, x), [level volumes float64, coarsest first])."""
        buf = np.frombuffer(stream, dtype=np.uint8)
        d = [_sz(0) for _ in range(3)]
        dst, nlev = _vp(None), _sz(0)
        ldims = ((_sz * 3) * 16)()
        levels = (_vp * 16)()
        rtn = self.lib.orc_decomp_3d_multi_res(buf.ctypes.data, _sz(buf.size), _sz(nthreads),
                                               C.byref(d[0]), C.byref(d[1]), C.byref(d[2]),
                                               C.byref(dst), C.byref(nlev), ldims, levels)
        if rtn:
            raise RuntimeError(f"orc_decomp_3d_multi_res returned {rtn}")
        shape = (d[2].value, d[1].value, d[0].value)
        vol = np.frombuffer(C.string_at(dst.value, int(np.prod(shape)) * 8), dtype=np.float64).reshape(shape).copy()
        self._libc.free(dst)
        out = []
        for h in range(nlev.value):
            sh = (ldims[h][2], ldims[h][1], ldims[h][0])
            out.append(np.frombuffer(C.string_at(levels[h], int(np.prod(sh)) * 8),
                                     dtype=np.float64).reshape(sh).copy())
            self._libc.free(_vp(levels[h]))
        return vol, out

    def speck1d_encode(self, coef, sign_bools):
        """coef: uint64[n]; sign_bools: bool[n] (True = non-negative). Returns the stream."""
        c = np.ascontiguousarray(coef, dtype=np.uint64)
        sign = pack_mask(sign_bools)
        out, n = _vp(None), _sz(0)
        self.lib.orc_speck1d_encode(c.ctypes.data, sign.ctypes.data, C.c_size_t(c.size),
                                    C.byref(out), C.byref(n))
        s = C.string_at(out.value, n.value)
        self._libc.free(out)
        return s

    def speck1d_decode(self, stream, n):
        buf = np.frombuffer(stream, dtype=np.uint8)
        coef = np.zeros(n, dtype=np.uint64)
        sign = np.zeros((n + 63) // 64, dtype=np.uint64)
        rtn = self.lib.orc_speck1d_decode(buf.ctypes.data, C.c_size_t(buf.size), C.c_size_t(n),
                                          coef.ctypes.data, sign.ctypes.data)
        if rtn:
            raise RuntimeError(f"orc_speck1d_decode returned {rtn}")
        return coef, unpack_mask(sign, n)

    def chunk_compress_rate(self, vol, bpp):
        v = np.array(vol, dtype=np.float64, order="C")
        dz, dy, dx = v.shape
        out, n = _vp(None), _sz(0)
        rtn = self.lib.orc_chunk_compress_rate(v.ctypes.data, self._dims((dx, dy, dz)), bpp,
                                               C.byref(out), C.byref(n))
        if rtn:
            raise RuntimeError(f"orc_chunk_compress_rate returned {rtn}")
        s = C.string_at(out.value, n.value)
        self._libc.free(out)
        return s

    def chunk_decompress(self, stream, shape_zyx):
        dz, dy, dx = shape_zyx
        buf = np.frombuffer(stream, dtype=np.uint8)
        out = np.zeros(shape_zyx, dtype=np.float64)
        rtn = self.lib.orc_chunk_decompress(buf.ctypes.data, buf.size, self._dims((dx, dy, dz)),
                                            out.ctypes.data)
        if rtn:
            raise RuntimeError(f"orc_chunk_decompress returned {rtn}")
        return out

    def chunk_volume(self, vol_xyz, chunk_xyz):
        n = self.lib.orc_chunk_volume(self._dims(vol_xyz), self._dims(chunk_xyz), None, 0)
        out = np.zeros((n, 6), dtype=np.uint64)
        self.lib.orc_chunk_volume(self._dims(vol_xyz), self._dims(chunk_xyz), out.ctypes.data, n)
        return out


class Ref(_CApiMixin):
    """The real reference (oracle/_ref, built from /root/reference by oracle/Makefile)."""

    def __init__(self):
        if not have_ref():
            raise FileNotFoundError("oracle/_ref is not built (run `make -C oracle ref`)")
        self.lib = C.CDLL(os.path.join(HERE, "_ref", "libSPERR_ref.so"), mode=C.RTLD_GLOBAL)
        self.probe = C.CDLL(os.path.join(HERE, "_ref", "libref_probe.so"))
        self._comp, self._decomp = self.lib.sperr_comp_3d, self.lib.sperr_decomp_3d
        self._trunc = self.lib.sperr_trunc_3d
        self._comp2, self._decomp2 = self.lib.sperr_comp_2d, self.lib.sperr_decomp_2d
        self._decomp2mr = self.probe.refp_decomp_2d_multi_res
        self._wire()
        P = self.probe
        P.refp_dwt3d.argtypes = [_vp, _sz, _sz, _sz]
        P.refp_idwt3d.argtypes = [_vp, _sz, _sz, _sz]
        P.refp_condition.argtypes = [_vp, _sz, _sz, _sz, _vp]
        P.refp_condition.restype = C.c_int
        P.refp_speck3d_encode.argtypes = [_vp, _vp, _sz, _sz, _sz, _sz, C.c_int, C.POINTER(_vp),
                                          C.POINTER(_sz)]
        P.refp_speck3d_encode.restype = C.c_int
        P.refp_speck3d_decode.argtypes = [_vp, _sz, _sz, _sz, _sz, _vp, _vp]
        P.refp_speck3d_decode.restype = C.c_int
        P.refp_speck1d_encode.argtypes = [_vp, _vp, _sz, C.c_int, C.POINTER(_vp), C.POINTER(_sz)]
        P.refp_speck1d_encode.restype = C.c_int
        P.refp_speck1d_decode.argtypes = [_vp, _sz, _sz, _vp, _vp]
        P.refp_speck1d_decode.restype = C.c_int
        P.refp_chunk_compress_rate.argtypes = [_vp, _sz, _sz, _sz, C.c_double, C.POINTER(_vp),
                                               C.POINTER(_sz)]
        P.refp_chunk_compress_rate.restype = C.c_int
        P.refp_chunk_decompress.argtypes = [_vp, _sz, _sz, _sz, _sz, _vp]
        P.refp_chunk_decompress.restype = C.c_int
        P.refp_chunk_volume.argtypes = [_sz] * 6 + [_vp, _sz]
        P.refp_chunk_volume.restype = _sz

    def dwt3d(self, vol):
        v = np.array(vol, dtype=np.float64, order="C")
        dz, dy, dx = v.shape
        self.probe.refp_dwt3d(v.ctypes.data, dx, dy, dz)
        return v

    def idwt3d(self, vol):
        v = np.array(vol, dtype=np.float64, order="C")
        dz, dy, dx = v.shape
        self.probe.refp_idwt3d(v.ctypes.data, dx, dy, dz)
        return v

    def condition(self, vol):
        v = np.array(vol, dtype=np.float64, order="C")
        dz, dy, dx = v.shape
        hdr = np.zeros(17, dtype=np.uint8)
        const = self.probe.refp_condition(v.ctypes.data, dx, dy, dz, hdr.ctypes.data)
        return v, hdr.tobytes(), bool(const)

    def speck3d_encode(self, coef, sign, budget_bits=0, width=8):
        c = np.ascontiguousarray(coef, dtype=np.uint64)
        dz, dy, dx = c.shape
        out, n = _vp(None), _sz(0)
        rtn = self.probe.refp_speck3d_encode(c.ctypes.data, sign.ctypes.data, dx, dy, dz,
                                             budget_bits, width, C.byref(out), C.byref(n))
        if rtn:
            raise RuntimeError(f"refp_speck3d_encode returned {rtn}")
        s = C.string_at(out.value, n.value)
        self._libc.free(out)
        return s

    def speck3d_decode(self, stream, shape_zyx):
        dz, dy, dx = shape_zyx
        buf = np.frombuffer(stream, dtype=np.uint8)
        coef = np.zeros(shape_zyx, dtype=np.uint64)
        sign = np.zeros((coef.size + 63) // 64, dtype=np.uint64)
        self.probe.refp_speck3d_decode(buf.ctypes.data, buf.size, dx, dy, dz, coef.ctypes.data,
                                       sign.ctypes.data)
        return coef, sign

    def decomp_3d_multi_res(self, stream, nthreads=1):
        buf = np.frombuffer(stream, dtype=np.uint8)
        dims = (_sz * 3)()
        vol = _vp(None)
        ldims = ((_sz * 3) * 16)()
        levels = (_vp * 16)()
        self.probe.refp_decomp_multi_res.restype = C.c_int
        self.probe.refp_decomp_multi_res.argtypes = [_vp, _sz, _sz, _vp, C.POINTER(_vp), _vp, _vp, _sz]
        n = self.probe.refp_decomp_multi_res(buf.ctypes.data, _sz(buf.size), _sz(nthreads), dims,
                                             C.byref(vol), ldims, levels, _sz(16))
        if n < 0:
            raise RuntimeError("refp_decomp_multi_res failed")
        shape = (dims[2], dims[1], dims[0])
        v = np.frombuffer(C.string_at(vol.value, int(np.prod(shape)) * 8), dtype=np.float64).reshape(shape).copy()
        self._libc.free(vol)
        out = []
        for h in range(n):
            sh = (ldims[h][2], ldims[h][1], ldims[h][0])
            out.append(np.frombuffer(C.string_at(levels[h], int(np.prod(sh)) * 8),
                                     dtype=np.float64).reshape(sh).copy())
            self._libc.free(_vp(levels[h]))
        return v, out

    def speck1d_encode(self, coef, sign_bools, width=8):
        c = np.ascontiguousarray(coef, dtype=np.uint64)
        sign = pack_mask(sign_bools)
        out, n = _vp(None), _sz(0)
        rtn = self.probe.refp_speck1d_encode(c.ctypes.data, sign.ctypes.data, c.size, width,
                                             C.byref(out), C.byref(n))
        if rtn:
            raise RuntimeError(f"refp_speck1d_encode returned {rtn}")
        s = C.string_at(out.value, n.value)
        self._libc.free(out)
        return s

    def speck1d_decode(self, stream, n):
        buf = np.frombuffer(stream, dtype=np.uint8)
        coef = np.zeros(n, dtype=np.uint64)
        sign = np.zeros((n + 63) // 64, dtype=np.uint64)
        self.probe.refp_speck1d_decode(buf.ctypes.data, buf.size, n, coef.ctypes.data,
                                       sign.ctypes.data)
        return coef, unpack_mask(sign, n)

    def chunk_compress_rate(self, vol, bpp):
        v = np.array(vol, dtype=np.float64, order="C")
        dz, dy, dx = v.shape
        out, n = _vp(None), _sz(0)
        rtn = self.probe.refp_chunk_compress_rate(v.ctypes.data, dx, dy, dz, bpp, C.byref(out),
                                                  C.byref(n))
        if rtn:
            raise RuntimeError(f"refp_chunk_compress_rate returned {rtn}")
        s = C.string_at(out.value, n.value)
        self._libc.free(out)
        return s

    def chunk_decompress(self, stream, shape_zyx):
        dz, dy, dx = shape_zyx
        buf = np.frombuffer(stream, dtype=np.uint8)
        out = np.zeros(shape_zyx, dtype=np.float64)
        rtn = self.probe.refp_chunk_decompress(buf.ctypes.data, buf.size, dx, dy, dz,
                                               out.ctypes.data)
        if rtn:
            raise RuntimeError(f"refp_chunk_decompress returned {rtn}")
        return out

    def chunk_volume(self, vol_xyz, chunk_xyz):
        n = self.probe.refp_chunk_volume(*vol_xyz, *chunk_xyz, None, 0)
        out = np.zeros((n, 6), dtype=np.uint64)
        self.probe.refp_chunk_volume(*vol_xyz, *chunk_xyz, out.ctypes.data, n)
        return out

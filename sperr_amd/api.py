"""Python binding of libsperr_hip.so (ctypes) -- plumbing for tests and bench.py.

torch is used only for device memory, streams and (in bench.py) torch.distributed; the compute
path is the C-ABI library declared in include/sperr_hip.h.  There is no CPU fallback: importing
works anywhere, but every call needs a GPU and raises if the library is missing or a call fails.
"""
import ctypes as C
import os

import numpy as np

# the decoder runs the shape groups of a volume side by side on up to 8 streams: ask the ROCm runtime
# for as many hardware queues (default 4; only read at the process's first HIP call)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

_HERE = os.path.dirname(os.path.abspath(__file__))
# (SPERR_HIP_LIB: another build of the same library, e.g. one with diagnostic switches compiled in)
LIB_PATH = os.environ.get("SPERR_HIP_LIB") or os.path.join(_HERE, "libsperr_hip.so")
_sz, _vp = C.c_size_t, C.c_void_p

# every symbol include/sperr_hip.h declares
EXPORTS = [
    "sperr_comp_3d", "sperr_decomp_3d", "sperr_parse_header", "sperr_trunc_3d",
    "sperrhip_max_compressed_size", "sperrhip_compress_dev", "sperrhip_decompress_dev",
    "sperrhip_parse_header_dev", "sperrhip_dwt3d_dev", "sperrhip_speck3d_encode_dev",
    "sperrhip_speck3d_decode_dev", "sperrhip_profile_enable", "sperrhip_profile_reset",
    "sperrhip_profile_get", "sperrhip_profile_get2", "sperrhip_profile_only",
    "sperrhip_multires_levels", "sperrhip_decompress_multires_dev", "sperrhip_decomp_3d_multires",
    "sperr_comp_2d", "sperr_decomp_2d", "sperrhip_max_compressed_size_2d", "sperrhip_compress_2d_dev",
    "sperrhip_decompress_2d_dev", "sperrhip_version", "sperrhip_debug_lis_stamps",
    "sperrhip_multires_levels_2d", "sperrhip_decompress_2d_multires_dev", "sperrhip_decomp_2d_multires",
    "sperrhip_comp_3d_farm", "sperrhip_decomp_3d_farm", "sperrhip_decomp_3d_into",
    "sperrhip_farm_selftest", "sperrhip_release", "sperrhip_debug_counter",
    "sperrhip_numa_probe", "sperrhip_numa_bind_self", "sperrhip_farm_device_place",
    "sperrhip_host_cpus", "sperrhip_host_throttle", "sperrhip_farm_threads",
]


class SperrHipError(RuntimeError):
    pass


def host_cpus(lib):
    """CPUs this process may use, as the library's farm sees them (sperr_amd/csrc/host_cpus.hpp): the machine's
    count, the affinity mask, the cgroup's CFS quota in CPUs (0.0: none) and the smallest of them."""
    vis, aff, use = _sz(0), _sz(0), _sz(0)
    quota = C.c_double(0.0)
    lib.sperrhip_host_cpus.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(_sz), C.POINTER(_sz), C.POINTER(C.c_double),
                                       C.POINTER(_sz)]
    rc = lib.sperrhip_host_cpus(None, None, C.byref(vis), C.byref(aff), C.byref(quota), C.byref(use))
    if rc != 0:
        raise SperrHipError(f"sperrhip_host_cpus returned {rc}")
    return {"cores_visible": vis.value, "affinity": aff.value, "cpu_quota": quota.value or None, "usable": use.value}


def host_throttle(lib):
    """(nr_throttled, throttled_usec) of the process's cgroup so far, or None when there is no cpu.stat"""
    nr, us = C.c_ulonglong(0), C.c_ulonglong(0)
    lib.sperrhip_host_throttle.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
    if lib.sperrhip_host_throttle(None, None, C.byref(nr), C.byref(us)) != 0:
        return None
    return nr.value, us.value


def load_library():
    if not os.path.exists(LIB_PATH):
        raise SperrHipError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # torch brings a HIP runtime of its own: when the library is loaded first it binds to the system's,
    # and a process with two runtimes does not see the GPU through the second one.  Callers here move
    # their data with torch anyway, so its runtime is loaded first and serves both.
    try:
        import torch  # noqa: F401
    except Exception:   # noqa: BLE001  (the loader itself does not need it)
        pass
    lib = C.CDLL(LIB_PATH)
    lib.sperr_comp_3d.restype = C.c_int
    lib.sperr_comp_3d.argtypes = [_vp, C.c_int, _sz, _sz, _sz, _sz, _sz, _sz, C.c_int, C.c_double,
                                  _sz, C.POINTER(_vp), C.POINTER(_sz)]
    lib.sperr_decomp_3d.restype = C.c_int
    lib.sperr_decomp_3d.argtypes = [_vp, _sz, C.c_int, _sz, C.POINTER(_sz), C.POINTER(_sz),
                                    C.POINTER(_sz), C.POINTER(_vp)]
    lib.sperr_parse_header.restype = None
    lib.sperr_parse_header.argtypes = [_vp, C.POINTER(_sz), C.POINTER(_sz), C.POINTER(_sz),
                                       C.POINTER(C.c_int)]
    lib.sperr_trunc_3d.restype = C.c_int
    lib.sperr_trunc_3d.argtypes = [_vp, _sz, C.c_uint, C.POINTER(_vp), C.POINTER(_sz)]
    lib.sperrhip_max_compressed_size.restype = _sz
    lib.sperrhip_max_compressed_size.argtypes = [_sz] * 6 + [C.c_int, C.c_double]
    lib.sperrhip_compress_dev.restype = C.c_int
    lib.sperrhip_compress_dev.argtypes = [_vp, C.c_int, _sz, _sz, _sz, _sz, _sz, _sz, C.c_int,
                                          C.c_double, _vp, _sz, C.POINTER(_sz), _vp]
    lib.sperrhip_decompress_dev.restype = C.c_int
    lib.sperrhip_decompress_dev.argtypes = [_vp, _sz, C.c_int, _vp, _sz, C.POINTER(_sz),
                                            C.POINTER(_sz), C.POINTER(_sz), _vp]
    lib.sperrhip_parse_header_dev.restype = C.c_int
    lib.sperrhip_parse_header_dev.argtypes = [_vp, _sz] + [C.POINTER(_sz)] * 3 + \
        [C.POINTER(C.c_int)] + [C.POINTER(_sz)] * 3
    lib.sperrhip_dwt3d_dev.restype = C.c_int
    lib.sperrhip_dwt3d_dev.argtypes = [_vp, _sz, _sz, _sz, C.c_int, _vp]
    lib.sperrhip_speck3d_encode_dev.restype = C.c_int
    lib.sperrhip_speck3d_encode_dev.argtypes = [_vp, C.c_int, _vp, _sz, _sz, _sz, _sz, _vp, _sz,
                                                C.POINTER(_sz), _vp]
    lib.sperrhip_speck3d_decode_dev.restype = C.c_int
    lib.sperrhip_speck3d_decode_dev.argtypes = [_vp, _sz, _sz, _sz, _sz, _vp, _vp,
                                                C.POINTER(C.c_int), _vp]
    lib.sperrhip_profile_enable.argtypes = [C.c_int]
    lib.sperrhip_profile_only.argtypes = [C.c_char_p]
    lib.sperrhip_profile_get2.restype = C.c_int
    lib.sperrhip_profile_get2.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_double),
                                          C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_int]
    lib.sperrhip_profile_get.restype = C.c_int
    lib.sperrhip_profile_get.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_double),
                                         C.POINTER(C.c_int), C.c_int]
    lib.sperrhip_version.restype = C.c_char_p
    lib.sperrhip_comp_3d_farm.restype = C.c_int
    lib.sperrhip_comp_3d_farm.argtypes = [_vp, C.c_int, _sz, _sz, _sz, _sz, _sz, _sz, C.c_int, C.c_double,
                                          _sz, _vp, _sz, C.POINTER(_vp), C.POINTER(_sz)]
    lib.sperrhip_decomp_3d_farm.restype = C.c_int
    lib.sperrhip_decomp_3d_farm.argtypes = [_vp, _sz, C.c_int, _sz, _vp, _sz, C.POINTER(_sz),
                                            C.POINTER(_sz), C.POINTER(_sz), C.POINTER(_vp)]
    lib.sperrhip_decomp_3d_into.restype = C.c_int
    lib.sperrhip_decomp_3d_into.argtypes = [_vp, _sz, C.c_int, _sz, _vp, _sz, _vp, _sz, C.POINTER(_sz),
                                            C.POINTER(_sz), C.POINTER(_sz)]
    return lib


class SperrHip:
    """Device-resident SPERR 3D compressor / decompressor (fixed-rate mode)."""

    def __init__(self):
        import torch
        self.torch = torch
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise SperrHipError("no GPU visible: libsperr_hip has no CPU fallback")
        self._libc = C.CDLL(None)
        self._libc.free.argtypes = [_vp]

    def _stream(self):
        return _vp(self.torch.cuda.current_stream().cuda_stream)

    # ---- device-resident API -------------------------------------------------------------
    def max_compressed_size(self, shape_zyx, chunks_xyz, quality, mode=1):
        dz, dy, dx = shape_zyx
        return self.lib.sperrhip_max_compressed_size(dx, dy, dz, *chunks_xyz, mode, quality)

    def compress(self, vol, chunks_xyz, bpp, out=None, mode=1):
        """vol: cuda tensor float32/float64 shaped (z, y, x). Returns a cuda uint8 tensor view of
        the container (a slice of `out` when given).  mode 1: `bpp` is the bit rate; mode 2: it is
        the target PSNR in dB; mode 3: the point-wise error tolerance (the reference's modes,
        include/SPERR_C_API.h:95-99)."""
        torch = self.torch
        assert vol.is_cuda and vol.is_contiguous() and vol.dim() == 3
        assert vol.dtype in (torch.float32, torch.float64)
        dz, dy, dx = vol.shape
        cap = self.max_compressed_size(vol.shape, chunks_xyz, bpp, mode)
        if out is None or out.numel() < cap:
            out = torch.empty(cap, dtype=torch.uint8, device=vol.device)
        n = _sz(0)
        rtn = self.lib.sperrhip_compress_dev(vol.data_ptr(), int(vol.dtype == torch.float32), dx,
                                             dy, dz, *chunks_xyz, mode, float(bpp), out.data_ptr(),
                                             out.numel(), C.byref(n), self._stream())
        if rtn != 0:
            raise SperrHipError(f"sperrhip_compress_dev returned {rtn}")
        return out[:n.value]

    def parse_header(self, container):
        d = [_sz(0) for _ in range(6)]
        isf = C.c_int(0)
        rtn = self.lib.sperrhip_parse_header_dev(container.data_ptr(), container.numel(),
                                                 C.byref(d[0]), C.byref(d[1]), C.byref(d[2]),
                                                 C.byref(isf), C.byref(d[3]), C.byref(d[4]),
                                                 C.byref(d[5]))
        if rtn != 0:
            raise SperrHipError(f"sperrhip_parse_header_dev returned {rtn}")
        return (d[2].value, d[1].value, d[0].value), bool(isf.value), \
               (d[3].value, d[4].value, d[5].value)

    def decompress(self, container, output_float=True, out=None, shape_zyx=None):
        torch = self.torch
        assert container.is_cuda and container.dtype == torch.uint8 and container.is_contiguous()
        if shape_zyx is None:
            shape_zyx, _, _ = self.parse_header(container)
        dt = torch.float32 if output_float else torch.float64
        if out is None:
            out = torch.empty(shape_zyx, dtype=dt, device=container.device)
        assert out.dtype == dt and out.is_contiguous()
        dx, dy, dz = _sz(0), _sz(0), _sz(0)
        rtn = self.lib.sperrhip_decompress_dev(container.data_ptr(), container.numel(),
                                               int(output_float), out.data_ptr(),
                                               out.numel() * out.element_size(), C.byref(dx),
                                               C.byref(dy), C.byref(dz), self._stream())
        if rtn != 0:
            raise SperrHipError(f"sperrhip_decompress_dev returned {rtn}")
        return out

    # ---- stage access ----------------------------------------------------------------------
    def dwt3d(self, vals, inverse=False):
        """In-place on a contiguous cuda float64 tensor shaped (z, y, x)."""
        torch = self.torch
        assert vals.is_cuda and vals.dtype == torch.float64 and vals.is_contiguous()
        dz, dy, dx = vals.shape
        rtn = self.lib.sperrhip_dwt3d_dev(vals.data_ptr(), dx, dy, dz, int(inverse), self._stream())
        if rtn != 0:
            raise SperrHipError(f"sperrhip_dwt3d_dev returned {rtn}")
        return vals

    def speck3d_encode(self, coef, sign, budget_bits=0):
        """coef: cuda int32 (uint32 bits) or int64 (uint64 bits) tensor (z, y, x); sign: cuda int64
        tensor of (n+63)//64 words. Returns the stream as bytes."""
        torch = self.torch
        assert coef.is_cuda and coef.is_contiguous() and sign.is_cuda
        width = coef.element_size()
        dz, dy, dx = coef.shape
        n = coef.numel()
        cap = 9 + (budget_bits + 7) // 8 + 64 if budget_bits else 9 + 70 * n // 8 + 4096
        out = torch.empty(cap, dtype=torch.uint8, device=coef.device)
        ln = _sz(0)
        rtn = self.lib.sperrhip_speck3d_encode_dev(coef.data_ptr(), width, sign.data_ptr(), dx, dy,
                                                   dz, budget_bits, out.data_ptr(), cap,
                                                   C.byref(ln), self._stream())
        if rtn != 0:
            raise SperrHipError(f"sperrhip_speck3d_encode_dev returned {rtn}")
        return bytes(out[:ln.value].cpu().numpy())

    def speck3d_decode(self, stream, shape_zyx):
        """Returns (coef uint64 numpy (z,y,x), sign uint64 numpy words)."""
        torch = self.torch
        dz, dy, dx = shape_zyx
        n = dz * dy * dx
        s = torch.from_numpy(np.frombuffer(stream, dtype=np.uint8).copy()).cuda()
        coef = torch.zeros(n, dtype=torch.int64, device="cuda")
        sign = torch.zeros((n + 63) // 64, dtype=torch.int64, device="cuda")
        w = C.c_int(0)
        rtn = self.lib.sperrhip_speck3d_decode_dev(s.data_ptr(), s.numel(), dx, dy, dz,
                                                   coef.data_ptr(), sign.data_ptr(), C.byref(w),
                                                   self._stream())
        if rtn != 0:
            raise SperrHipError(f"sperrhip_speck3d_decode_dev returned {rtn}")
        raw = coef.cpu().numpy()
        if w.value == 4:
            c = raw.view(np.uint32)[:n].astype(np.uint64)
        else:
            c = raw.view(np.uint64)
        return c.reshape(shape_zyx), sign.cpu().numpy().view(np.uint64)

    # ---- reference-compatible host API (include/sperr_hip.h) ------------------------------
    def comp_3d(self, vol, chunks_xyz, mode, quality, nthreads=0):
        vol = np.ascontiguousarray(vol)
        dz, dy, dx = vol.shape
        dst, n = _vp(None), _sz(0)
        rtn = self.lib.sperr_comp_3d(vol.ctypes.data, int(vol.dtype == np.float32), dx, dy, dz,
                                     *chunks_xyz, mode, quality, nthreads, C.byref(dst),
                                     C.byref(n))
        if rtn != 0:
            raise SperrHipError(f"sperr_comp_3d returned {rtn}")
        out = C.string_at(dst.value, n.value)
        self._libc.free(dst)
        return out

    def trunc_3d(self, stream, pct):
        buf = np.frombuffer(stream, dtype=np.uint8)
        dst, n = _vp(None), _sz(0)
        rtn = self.lib.sperr_trunc_3d(buf.ctypes.data, buf.size, pct, C.byref(dst), C.byref(n))
        if rtn != 0:
            raise SperrHipError(f"sperr_trunc_3d returned {rtn}")
        out = C.string_at(dst.value, n.value)
        self._libc.free(dst)
        return out

    def decomp_3d(self, stream, output_float=True, nthreads=0):
        buf = np.frombuffer(stream, dtype=np.uint8)
        dst = _vp(None)
        dx, dy, dz = _sz(0), _sz(0), _sz(0)
        rtn = self.lib.sperr_decomp_3d(buf.ctypes.data, buf.size, int(output_float), nthreads,
                                       C.byref(dx), C.byref(dy), C.byref(dz), C.byref(dst))
        if rtn != 0:
            raise SperrHipError(f"sperr_decomp_3d returned {rtn}")
        n = dx.value * dy.value * dz.value
        dt = np.float32 if output_float else np.float64
        out = np.frombuffer(C.string_at(dst.value, n * np.dtype(dt).itemsize), dtype=dt).copy()
        self._libc.free(dst)
        return out.reshape(dz.value, dy.value, dx.value)

    # ---- chunk farm on an explicit device list (include/sperr_hip.h) -------------------------
    @staticmethod
    def _devs(devices):
        if not devices:
            return None, 0
        arr = (C.c_int * len(devices))(*devices)
        return arr, len(devices)

    def comp_3d_farm(self, vol, chunks_xyz, mode, quality, devices=None, nthreads=0):
        """vol: numpy (z, y, x) or a pinned torch CPU tensor; -> container bytes."""
        ptr, shape, is_float = self._host_array(vol)
        dz, dy, dx = shape
        dst, n = _vp(None), _sz(0)
        arr, nd = self._devs(devices)
        rtn = self.lib.sperrhip_comp_3d_farm(ptr, is_float, dx, dy, dz, *chunks_xyz, mode, quality,
                                             nthreads, arr, nd, C.byref(dst), C.byref(n))
        if rtn != 0:
            raise SperrHipError(f"sperrhip_comp_3d_farm returned {rtn}")
        out = C.string_at(dst.value, n.value)
        self._libc.free(dst)
        return out

    def _host_array(self, vol, writable=False):
        """(pointer, shape, is_float) of a host volume.  float32 or float64 only -- anything else would
        be read with the wrong element size; an output buffer must be C-contiguous (a copy made here
        would take the values and the caller's array would stay empty)."""
        if isinstance(vol, np.ndarray):
            if vol.dtype not in (np.float32, np.float64):
                raise SperrHipError(f"host volumes are float32 or float64, not {vol.dtype}")
            if writable and not (vol.flags.c_contiguous and vol.flags.writeable):
                raise SperrHipError("the output array must be C-contiguous and writable")
            vol = np.ascontiguousarray(vol)
            self._keep = vol
            return vol.ctypes.data, vol.shape, int(vol.dtype == np.float32)
        if vol.is_cuda or not vol.is_contiguous():
            raise SperrHipError("a torch host volume must be a contiguous CPU tensor")
        if vol.dtype not in (self.torch.float32, self.torch.float64):
            raise SperrHipError(f"host volumes are float32 or float64, not {vol.dtype}")
        return vol.data_ptr(), tuple(vol.shape), int(vol.dtype == self.torch.float32)

    def release(self):
        """sperrhip_release: idle engines and farm workers give their memory back."""
        self.lib.sperrhip_release.restype = None
        self.lib.sperrhip_release.argtypes = []
        self.lib.sperrhip_release()

    def decomp_3d_farm(self, stream, output_float=True, devices=None, nthreads=0):
        buf = np.frombuffer(stream, dtype=np.uint8)
        dst = _vp(None)
        dx, dy, dz = _sz(0), _sz(0), _sz(0)
        arr, nd = self._devs(devices)
        rtn = self.lib.sperrhip_decomp_3d_farm(buf.ctypes.data, buf.size, int(output_float), nthreads,
                                               arr, nd, C.byref(dx), C.byref(dy), C.byref(dz),
                                               C.byref(dst))
        if rtn != 0:
            raise SperrHipError(f"sperrhip_decomp_3d_farm returned {rtn}")
        n = dx.value * dy.value * dz.value
        dt = np.float32 if output_float else np.float64
        out = np.frombuffer(C.string_at(dst.value, n * np.dtype(dt).itemsize), dtype=dt).copy()
        self._libc.free(dst)
        return out.reshape(dz.value, dy.value, dx.value)

    def decomp_3d_into(self, stream, out, devices=None, nthreads=0):
        """out: numpy array or (pinned) torch CPU tensor shaped (z, y, x), float32 or float64."""
        buf = np.frombuffer(stream, dtype=np.uint8) if isinstance(stream, (bytes, bytearray)) else stream
        sptr = buf.ctypes.data if isinstance(buf, np.ndarray) else buf.data_ptr()
        slen = buf.size if isinstance(buf, np.ndarray) else buf.numel()
        ptr, shape, is_float = self._host_array(out, writable=True)
        nbytes = int(np.prod(shape)) * (4 if is_float else 8)
        dx, dy, dz = _sz(0), _sz(0), _sz(0)
        arr, nd = self._devs(devices)
        rtn = self.lib.sperrhip_decomp_3d_into(sptr, slen, is_float, nthreads, arr, nd, ptr, nbytes,
                                               C.byref(dx), C.byref(dy), C.byref(dz))
        if rtn != 0:
            raise SperrHipError(f"sperrhip_decomp_3d_into returned {rtn}")
        assert (dz.value, dy.value, dx.value) == tuple(shape)
        return out

    # ---- 2D slices --------------------------------------------------------------------------
    def compress_2d(self, img, quality, mode=1, header=False):
        """img: cuda tensor float32/float64 (y, x). Returns a cuda uint8 tensor (sperr_comp_2d)."""
        torch = self.torch
        assert img.is_cuda and img.is_contiguous() and img.dim() == 2
        dy, dx = img.shape
        self.lib.sperrhip_max_compressed_size_2d.restype = _sz
        self.lib.sperrhip_max_compressed_size_2d.argtypes = [_sz, _sz, C.c_int, C.c_double]
        cap = self.lib.sperrhip_max_compressed_size_2d(dx, dy, mode, float(quality))
        out = torch.empty(cap, dtype=torch.uint8, device=img.device)
        n = _sz(0)
        self.lib.sperrhip_compress_2d_dev.argtypes = [C.c_void_p, C.c_int, _sz, _sz, C.c_int, C.c_double,
                                                      C.c_int, C.c_void_p, _sz, C.POINTER(_sz), C.c_void_p]
        rtn = self.lib.sperrhip_compress_2d_dev(img.data_ptr(), int(img.dtype == torch.float32), dx, dy,
                                                mode, float(quality), int(header), out.data_ptr(),
                                                out.numel(), C.byref(n), self._stream())
        if rtn != 0:
            raise SperrHipError(f"sperrhip_compress_2d_dev returned {rtn}")
        return out[:n.value]

    def decompress_2d(self, stream, shape_yx, output_float=True):
        """stream: cuda uint8 tensor WITHOUT the optional 10-byte header (sperr_decomp_2d)."""
        torch = self.torch
        dy, dx = shape_yx
        stream = stream.contiguous()
        out = torch.empty((dy, dx), dtype=torch.float32 if output_float else torch.float64,
                          device=stream.device)
        self.lib.sperrhip_decompress_2d_dev.argtypes = [C.c_void_p, _sz, C.c_int, _sz, _sz, C.c_void_p,
                                                        _sz, C.c_void_p]
        rtn = self.lib.sperrhip_decompress_2d_dev(stream.data_ptr(), stream.numel(), int(output_float),
                                                  dx, dy, out.data_ptr(),
                                                  out.numel() * out.element_size(), self._stream())
        if rtn != 0:
            raise SperrHipError(f"sperrhip_decompress_2d_dev returned {rtn}")
        return out

    # ---- multi-resolution decoding ----------------------------------------------------------
    def multires_levels(self, shape_zyx, chunks_xyz):
        """[(z, y, x) of every coarsened level], coarsest first (empty: no hierarchy exists)."""
        dz, dy, dx = shape_zyx
        nlev = _sz(0)
        dims = (_sz * 48)()
        self.lib.sperrhip_multires_levels.argtypes = [_sz] * 6 + [C.POINTER(_sz), C.POINTER(_sz)]
        if self.lib.sperrhip_multires_levels(dx, dy, dz, *chunks_xyz, C.byref(nlev), dims) != 0:
            raise SperrHipError("sperrhip_multires_levels failed")
        return [(dims[3 * h + 2], dims[3 * h + 1], dims[3 * h]) for h in range(nlev.value)]

    def decompress_multires(self, stream, output_float=True):
        """-> (volume, [float64 level volumes, coarsest first]); SPERR3D_OMP_D::decompress(p, true)."""
        torch = self.torch
        shape, _, chunks = self.parse_header(stream)   # (z, y, x), is_float, (x, y, z)
        lv = self.multires_levels(shape, chunks)
        out = torch.empty(shape, dtype=torch.float32 if output_float else torch.float64,
                          device=stream.device)
        levels = [torch.empty(s, dtype=torch.float64, device=stream.device) for s in lv]
        ptrs = (C.c_void_p * max(1, len(lv)))(*[t.data_ptr() for t in levels])
        self.lib.sperrhip_decompress_multires_dev.argtypes = [C.c_void_p, _sz, C.c_int, C.c_void_p,
                                                              _sz, _sz, C.c_void_p, C.c_void_p]
        rtn = self.lib.sperrhip_decompress_multires_dev(stream.data_ptr(), stream.numel(),
                                                        int(output_float), out.data_ptr(),
                                                        out.numel() * out.element_size(), len(lv),
                                                        ptrs, self._stream())
        if rtn != 0:
            raise SperrHipError(f"sperrhip_decompress_multires_dev returned {rtn}")
        return out, levels

    def multires_levels_2d(self, shape_yx):
        """[(y, x) of every coarsened level of a slice], coarsest first."""
        dy, dx = shape_yx
        nlev = _sz(0)
        dims = (_sz * 32)()
        self.lib.sperrhip_multires_levels_2d.argtypes = [_sz, _sz, C.POINTER(_sz), C.POINTER(_sz)]
        if self.lib.sperrhip_multires_levels_2d(dx, dy, C.byref(nlev), dims) != 0:
            raise SperrHipError("sperrhip_multires_levels_2d failed")
        return [(dims[2 * h + 1], dims[2 * h]) for h in range(nlev.value)]

    def decompress_2d_multires(self, stream, shape_yx, output_float=True):
        """-> (slice, [float64 level slices, coarsest first]); SPECK2D_FLT::decompress(true)."""
        torch = self.torch
        dy, dx = shape_yx
        stream = stream.contiguous()
        lv = self.multires_levels_2d(shape_yx)
        out = torch.empty((dy, dx), dtype=torch.float32 if output_float else torch.float64,
                          device=stream.device)
        levels = [torch.empty(s, dtype=torch.float64, device=stream.device) for s in lv]
        ptrs = (C.c_void_p * max(1, len(lv)))(*[t.data_ptr() for t in levels])
        self.lib.sperrhip_decompress_2d_multires_dev.argtypes = [C.c_void_p, _sz, C.c_int, _sz, _sz, C.c_void_p,
                                                                 _sz, _sz, C.c_void_p, C.c_void_p]
        rtn = self.lib.sperrhip_decompress_2d_multires_dev(stream.data_ptr(), stream.numel(), int(output_float),
                                                           dx, dy, out.data_ptr(),
                                                           out.numel() * out.element_size(), len(lv), ptrs,
                                                           self._stream())
        if rtn != 0:
            raise SperrHipError(f"sperrhip_decompress_2d_multires_dev returned {rtn}")
        return out, levels

    # ---- profiling ------------------------------------------------------------------------
    def profile(self, on=True, only=None):
        """Bracket kernel launches with HIP events; `only`: just that kernel (cheap)."""
        self.lib.sperrhip_profile_only(only.encode() if only else None)
        self.lib.sperrhip_profile_enable(int(on))
        if on:
            self.lib.sperrhip_profile_reset()

    def profile_report(self, with_sum=False):
        """{kernel: (ms during which it was running, launches)}; with_sum: (busy ms, launches, sum
        of the launch durations) -- launches of sub-batches on different streams overlap."""
        cap = 128
        names = (C.c_char_p * cap)()
        ms = (C.c_double * cap)()
        sm = (C.c_double * cap)()
        cnt = (C.c_int * cap)()
        n = self.lib.sperrhip_profile_get2(names, ms, sm, cnt, cap)
        if with_sum:
            return {names[i].decode(): (ms[i], cnt[i], sm[i]) for i in range(min(n, cap))}
        return {names[i].decode(): (ms[i], cnt[i]) for i in range(min(n, cap))}

"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol that
include/sperr_hip.h declares (no compute call is made: there is no GPU here)."""
import ctypes
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from sperr_amd import api
    if not os.path.exists(api.LIB_PATH):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "sperr_amd", "csrc"), "-j4"])
    lib = ctypes.CDLL(api.LIB_PATH)
    header = open(os.path.join(ROOT, "include", "sperr_hip.h")).read()
    declared = set(re.findall(r"\b(sperr(?:hip)?_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations found"
    assert declared == set(api.EXPORTS)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} is declared in include/sperr_hip.h but not exported"


def test_cpp_mirror_header_compiles():
    """include/sperr_hip.hpp (SPERR3D_OMP_C / _D, SPECK3D_FLT, SPECK2D_FLT, SPERR3D_Stream_Tools
    mirrors) is valid C++17; tests/cpp/mirror_check.cpp, which drives all of them, compiles."""
    src = '#include "sperr_hip.hpp"\nint main(){ sperr::SPERR3D_OMP_C c; c.set_bitrate(2.0); ' \
          'sperr::SPERR3D_OMP_D d; sperr::SPECK3D_FLT f; f.set_dims({8,8,8}); ' \
          'sperr::SPECK2D_FLT g; g.set_dims({8,8,1}); sperr::SPERR3D_Stream_Tools t; ' \
          'sperr::SPERR3D_Header h; (void)t; (void)h; return 0; }\n'
    subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I", os.path.join(ROOT, "include"),
                    "-x", "c++", "-"], input=src.encode(), check=True)
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I",
                    os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "mirror_check.cpp")],
                   check=True)


def test_no_gpu_means_loud_failure():
    """Without a GPU every entry point must fail loudly rather than fall back to a CPU path."""
    import numpy as np
    import torch
    if torch.cuda.is_available():
        return
    from sperr_amd import api
    lib = api.load_library()
    v = np.zeros((8, 8, 8), dtype=np.float32)
    dst, n = ctypes.c_void_p(None), ctypes.c_size_t(0)
    rtn = lib.sperr_comp_3d(v.ctypes.data, 1, 8, 8, 8, 8, 8, 8, 1, 2.0, 0, ctypes.byref(dst),
                            ctypes.byref(n))
    assert rtn == -1 and not dst.value


def test_truncation_refuses_headers_naming_more_chunks_than_bytes():
    """sperr_trunc_3d is host only: a header whose dimensions imply billions of chunks has to be
    refused (return code), not run into an allocation failure inside the library."""
    import numpy as np
    from sperr_amd import api
    lib = api.load_library()
    for vol, chunk in (((0x7fffffff,) * 3, (1, 1, 1)), ((0xffffffff,) * 3, (2, 3, 1))):
        head = bytes([0, 0x40 | 0x20 | 0x10]) + np.array(vol, dtype=np.uint32).tobytes() + \
            np.array(chunk, dtype=np.uint16).tobytes() + bytes(200)
        dst, n = ctypes.c_void_p(None), ctypes.c_size_t(0)
        rtn = lib.sperr_trunc_3d(head, ctypes.c_size_t(len(head)), 50, ctypes.byref(dst), ctypes.byref(n))
        assert rtn != 0 and not dst.value

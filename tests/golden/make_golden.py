"""Generate the golden vectors in tests/golden/ from the REAL reference (oracle/_ref, compiled
from /root/reference by oracle/Makefile).  Run in the dev container only:

    python tests/golden/make_golden.py

Inputs: two small files the reference's own tests use (test_data/wmag17.float, a crop of
test_data/vorticity.128_128_41, test_data/const32x20x16.float) copied as data fixtures, plus the
machine-independent integer-arithmetic field from tests/fields.py.  Expected outputs: the exact
container bytes produced by the reference's sperr_comp_3d and the SHA-256 of the floats/doubles
its sperr_decomp_3d returns, for several chunkings and bit rates (mode 1), target PSNRs (mode 2) and point-wise error
tolerances (mode 3).
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))

from fields import smooth_field  # noqa: E402
from oracle.pyoracle import Ref  # noqa: E402

TD = "/root/reference/test_data"


def main():
    ref = Ref()
    inputs = {}
    w17 = np.fromfile(f"{TD}/wmag17.float", dtype=np.float32).reshape(17, 17, 17)
    inputs["wmag17"] = w17
    vort = np.fromfile(f"{TD}/vorticity.128_128_41", dtype=np.float32).reshape(41, 128, 128)
    inputs["vort_crop"] = np.ascontiguousarray(vort[3:36, 40:76, 50:90])   # z33 y36 x40
    inputs["const32x20x16"] = np.fromfile(f"{TD}/const32x20x16.float",
                                          dtype=np.float32).reshape(16, 20, 32)
    for name, arr in inputs.items():
        arr.tofile(os.path.join(HERE, f"{name}.f32"))

    gen = {
        "smooth48": ((48, 48, 48), np.float32),
        "smooth_odd": ((19, 37, 50), np.float32),
        "smooth_f64": ((24, 40, 40), np.float64),
        "smooth_flat_z": ((9, 64, 64), np.float32),   # wavelet-packet (z has 1 level, xy have 3)
    }
    cases = []
    plan = [
        ("wmag17", (17, 17, 17), [0.5, 2.0, 4.0, 24.0]),
        ("vort_crop", (40, 36, 33), [1.0, 4.0]),
        ("vort_crop", (20, 18, 16), [2.0]),           # 2x2x2 chunks + merged remainders
        ("const32x20x16", (32, 20, 16), [2.0]),
        ("const32x20x16", (16, 10, 8), [2.0]),
        ("smooth48", (48, 48, 48), [0.25, 2.0, 8.0]),
        ("smooth48", (24, 24, 24), [2.0]),
        ("smooth_odd", (50, 37, 19), [2.0, 6.0]),
        ("smooth_f64", (40, 40, 24), [2.0, 40.0]),    # 40 bpp forces the high-precision retry
        ("smooth_flat_z", (64, 64, 9), [2.0]),
    ]
    # mode 2: the quality is a target PSNR in dB (include/SPERR_C_API.h:95-99)
    plan_psnr = [
        ("wmag17", (17, 17, 17), [60.0, 110.0]),
        ("vort_crop", (20, 18, 16), [80.0]),
        ("const32x20x16", (16, 10, 8), [90.0]),
        ("smooth_odd", (50, 37, 19), [70.0]),
        ("smooth_f64", (40, 40, 24), [100.0, 225.0]),   # 225 dB needs 64-bit coefficients
    ]
    # mode 3: the quality is the largest point-wise error allowed; the chunk streams carry an
    # outlier stream when the quantiser alone does not meet it (src/SPECK_FLT.cpp:461-486)
    plan_pwe = [
        ("wmag17", (17, 17, 17), [1.0, 0.001]),
        ("vort_crop", (20, 18, 16), [1e-3, 1e-5, 1e-7]),   # 1e-3: every coefficient quantises to 0
        ("const32x20x16", (16, 10, 8), [0.1]),
        ("smooth_odd", (50, 37, 19), [0.01]),
        ("smooth_f64", (40, 40, 24), [1e-9, 1e-13]),      # 64-bit coefficients, 5 outlier planes
    ]
    jobs = [(n, c, 1, q) for n, c, qs in plan for q in qs] + \
           [(n, c, 2, q) for n, c, qs in plan_psnr for q in qs] + \
           [(n, c, 3, q) for n, c, qs in plan_pwe for q in qs]
    for name, chunks, mode, bpp in jobs:
        if name in inputs:
            arr = inputs[name]
        else:
            shape, dt = gen[name]
            arr = smooth_field(shape, dtype=dt)
        if True:
            stream = ref.comp_3d(arr, chunks, mode, bpp)
            dec_f = ref.decomp_3d(stream, True)
            dec_d = ref.decomp_3d(stream, False)
            tag = f"{name}_c{chunks[0]}x{chunks[1]}x{chunks[2]}_{('bpp', 'psnr', 'pwe')[mode - 1]}{bpp}"
            with open(os.path.join(HERE, tag + ".sperr"), "wb") as f:
                f.write(stream)
            cases.append({
                "tag": tag, "input": name, "shape_zyx": list(arr.shape),
                "dtype": str(arr.dtype), "chunks_xyz": list(chunks), "mode": mode, "bpp": bpp,
                "input_sha256": hashlib.sha256(arr.tobytes()).hexdigest(),
                "stream_len": len(stream),
                "stream_sha256": hashlib.sha256(stream).hexdigest(),
                "decoded_f32_sha256": hashlib.sha256(dec_f.tobytes()).hexdigest(),
                "decoded_f64_sha256": hashlib.sha256(dec_d.tobytes()).hexdigest(),
            })
            print(tag, len(stream))
    # 2D slices through sperr_comp_2d / sperr_decomp_2d (SPECK2D_FLT): a crop of the reference's
    # test_data/999x999.float (BASELINE configs[3]: PSNR = 90 dB) and a generated field
    img = np.fromfile(f"{TD}/999x999.float", dtype=np.float32).reshape(999, 999)
    crop = np.ascontiguousarray(img[200:296, 300:421])   # y96 x121
    crop.tofile(os.path.join(HERE, "img999_crop.f32"))
    slices = {"img999_crop": crop, "smooth2d_f64": smooth_field((1, 50, 72), dtype=np.float64)[0]}
    cases2d = []
    for name, mode, q, hdr in [("img999_crop", 2, 90.0, True), ("img999_crop", 1, 2.0, False),
                               ("img999_crop", 3, 0.05, True), ("smooth2d_f64", 1, 6.0, False),
                               ("smooth2d_f64", 2, 120.0, True), ("smooth2d_f64", 3, 1e-6, False)]:
        arr = slices[name]
        stream = ref.comp_2d(arr, mode, q, hdr)
        body = stream[10:] if hdr else stream
        dec_f = ref.decomp_2d(body, arr.shape, True)
        dec_d = ref.decomp_2d(body, arr.shape, False)
        tag = f"{name}_2d_{('bpp', 'psnr', 'pwe')[mode - 1]}{q}{'_hdr' if hdr else ''}"
        with open(os.path.join(HERE, tag + ".sperr"), "wb") as f:
            f.write(stream)
        cases2d.append({
            "tag": tag, "input": name, "shape_yx": list(arr.shape), "dtype": str(arr.dtype),
            "mode": mode, "quality": q, "header": hdr,
            "input_sha256": hashlib.sha256(arr.tobytes()).hexdigest(),
            "stream_len": len(stream), "stream_sha256": hashlib.sha256(stream).hexdigest(),
            "decoded_f32_sha256": hashlib.sha256(dec_f.tobytes()).hexdigest(),
            "decoded_f64_sha256": hashlib.sha256(dec_d.tobytes()).hexdigest(),
        })
        print(tag, len(stream))
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_golden.py",
                   "reference": "NCAR/SPERR v0.8.5, g++ -O3 -mavx2 -mfma (oracle/Makefile)",
                   "cases": cases, "cases_2d": cases2d}, f, indent=1)


if __name__ == "__main__":
    main()

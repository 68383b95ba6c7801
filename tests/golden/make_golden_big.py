"""Golden vectors of the BASELINE.json configurations at (or near) full size, from the REAL
reference (oracle/_ref).  Run in the dev container only:

    python tests/golden/make_golden_big.py

Only hashes of what the reference produces are kept (container SHA-256 + length, SHA-256 of the
decoded floats / doubles); the inputs are two data files of the reference's own tests copied as data
fixtures (test_data/wmag91.float -- odd, not dyadic-friendly, SURVEY 8c -- and
test_data/999x999.float, BASELINE config 4) and fields from tests/fields.py whose bytes are the same
on every machine (the substitutes SURVEY 8d names for the missing wmag128.float of config 1 and
density_128x128x256.d64 of config 2).
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


def sha(b):
    return hashlib.sha256(b).hexdigest()


def main():
    ref = Ref()
    w91 = np.fromfile(f"{TD}/wmag91.float", dtype=np.float32).reshape(91, 91, 91)
    w91.tofile(os.path.join(HERE, "wmag91.f32"))
    img = np.fromfile(f"{TD}/999x999.float", dtype=np.float32).reshape(999, 999)
    img.tofile(os.path.join(HERE, "img999.f32"))
    vols = {
        "wmag91": w91,
        "smooth128": smooth_field((128, 128, 128), dtype=np.float32),          # config 1 substitute
        "smooth_f64_128x128x256": smooth_field((256, 128, 128), dtype=np.float64),   # config 2 substitute
    }
    plan = [
        ("wmag91", (91, 91, 91), 1, 2.0), ("wmag91", (91, 91, 91), 1, 4.0), ("wmag91", (64, 64, 64), 1, 4.0),
        ("wmag91", (64, 64, 64), 2, 100.0), ("wmag91", (32, 40, 50), 3, 1e-2),
        ("smooth128", (128, 128, 128), 1, 4.0),                                 # config 1: single chunk, BPP 4
        ("smooth_f64_128x128x256", (128, 128, 128), 3, 1e-6),                   # config 2: PWE 1e-6, 2 chunks
    ]
    cases = []
    for name, chunks, mode, q in plan:
        arr = vols[name]
        stream = ref.comp_3d(arr, chunks, mode, q, nthreads=8)
        dec_f = ref.decomp_3d(stream, True, nthreads=8)
        dec_d = ref.decomp_3d(stream, False, nthreads=8)
        tag = f"{name}_c{chunks[0]}x{chunks[1]}x{chunks[2]}_{('bpp', 'psnr', 'pwe')[mode - 1]}{q}"
        cases.append({"tag": tag, "input": name, "shape_zyx": list(arr.shape), "dtype": str(arr.dtype),
                      "chunks_xyz": list(chunks), "mode": mode, "quality": q,
                      "input_sha256": sha(arr.tobytes()), "stream_len": len(stream),
                      "stream_sha256": sha(stream), "decoded_f32_sha256": sha(dec_f.tobytes()),
                      "decoded_f64_sha256": sha(dec_d.tobytes())})
        print(tag, len(stream))
    cases2d = []
    for mode, q, hdr in [(2, 90.0, True), (1, 2.0, False), (3, 1e-3, True)]:   # config 4: PSNR 90
        stream = ref.comp_2d(img, mode, q, hdr)
        body = stream[10:] if hdr else stream
        dec_f = ref.decomp_2d(body, img.shape, True)
        dec_d = ref.decomp_2d(body, img.shape, False)
        tag = f"img999_2d_{('bpp', 'psnr', 'pwe')[mode - 1]}{q}{'_hdr' if hdr else ''}"
        cases2d.append({"tag": tag, "input": "img999", "shape_yx": [999, 999], "dtype": "float32",
                        "mode": mode, "quality": q, "header": hdr, "input_sha256": sha(img.tobytes()),
                        "stream_len": len(stream), "stream_sha256": sha(stream),
                        "decoded_f32_sha256": sha(dec_f.tobytes()), "decoded_f64_sha256": sha(dec_d.tobytes())})
        print(tag, len(stream))
    with open(os.path.join(HERE, "golden_big.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_golden_big.py",
                   "reference": "NCAR/SPERR v0.8.5, g++ -O3 -mavx2 -mfma (oracle/Makefile)",
                   "cases": cases, "cases_2d": cases2d}, f, indent=1)


if __name__ == "__main__":
    main()

"""BASELINE.json's configurations at (or near) full size against hashes of what the REAL reference
produces (tests/golden/golden_big.json, made by tests/golden/make_golden_big.py): wmag91.float
(SURVEY 8c), a 128^3 single chunk at BPP 4 (config 1), an fp64 128 x 128 x 256 volume in 128^3 chunks
at PWE 1e-6 (config 2), 999x999.float through the 2D path at PSNR 90 (config 4).  The oracle is
checked on the CPU, the HIP path on the GPU; both must reproduce the reference's container byte for
byte (SHA-256 + length) and its decoded floats and doubles bit for bit."""
import hashlib
import json
import os

import numpy as np
import pytest

from fields import smooth_field

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
with open(os.path.join(GOLD, "golden_big.json")) as f:
    _G = json.load(f)
CASES, CASES_2D = _G["cases"], _G["cases_2d"]


def sha(b):
    return hashlib.sha256(b).hexdigest()


def load(case):
    shape = tuple(case.get("shape_zyx") or case["shape_yx"])
    path = os.path.join(GOLD, case["input"] + ".f32")
    if os.path.exists(path):
        arr = np.fromfile(path, dtype=np.float32).reshape(shape)
    else:
        arr = smooth_field(shape, dtype=np.dtype(case["dtype"]))
    assert sha(arr.tobytes()) == case["input_sha256"]
    return arr


@pytest.mark.parametrize("case", CASES, ids=[c["tag"] for c in CASES])
def test_oracle_reproduces_the_reference(oracle, case):
    arr = load(case)
    got = oracle.comp_3d(arr, case["chunks_xyz"], case["mode"], case["quality"])
    assert len(got) == case["stream_len"] and sha(got) == case["stream_sha256"]
    assert sha(oracle.decomp_3d(got, True).tobytes()) == case["decoded_f32_sha256"]
    assert sha(oracle.decomp_3d(got, False).tobytes()) == case["decoded_f64_sha256"]


@pytest.mark.parametrize("case", CASES_2D, ids=[c["tag"] for c in CASES_2D])
def test_oracle_reproduces_the_reference_2d(oracle, case):
    img = load(case)
    got = oracle.comp_2d(img, case["mode"], case["quality"], case["header"])
    assert len(got) == case["stream_len"] and sha(got) == case["stream_sha256"]
    body = got[10:] if case["header"] else got
    assert sha(oracle.decomp_2d(body, img.shape, True).tobytes()) == case["decoded_f32_sha256"]
    assert sha(oracle.decomp_2d(body, img.shape, False).tobytes()) == case["decoded_f64_sha256"]


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=[c["tag"] for c in CASES])
def test_hip_reproduces_the_reference(case):
    import torch
    from sperr_amd.api import SperrHip
    eng = SperrHip()
    arr = load(case)
    dv = torch.from_numpy(arr).cuda()
    stream = eng.compress(dv, case["chunks_xyz"], case["quality"], mode=case["mode"])
    got = bytes(stream.cpu().numpy())
    assert len(got) == case["stream_len"] and sha(got) == case["stream_sha256"]
    assert sha(eng.decompress(stream, True).cpu().numpy().tobytes()) == case["decoded_f32_sha256"]
    assert sha(eng.decompress(stream, False).cpu().numpy().tobytes()) == case["decoded_f64_sha256"]
    # and through the reference-compatible host API (the chunk farm)
    assert sha(eng.comp_3d(arr, case["chunks_xyz"], case["mode"], case["quality"])) == case["stream_sha256"]
    assert sha(eng.decomp_3d(got, True).tobytes()) == case["decoded_f32_sha256"]


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES_2D, ids=[c["tag"] for c in CASES_2D])
def test_hip_reproduces_the_reference_2d(case):
    import torch
    from sperr_amd.api import SperrHip
    eng = SperrHip()
    img = load(case)
    stream = eng.compress_2d(torch.from_numpy(img).cuda(), case["quality"], mode=case["mode"], header=case["header"])
    got = bytes(stream.cpu().numpy())
    assert len(got) == case["stream_len"] and sha(got) == case["stream_sha256"]
    body = stream[10:] if case["header"] else stream
    assert sha(eng.decompress_2d(body, img.shape, True).cpu().numpy().tobytes()) == case["decoded_f32_sha256"]
    assert sha(eng.decompress_2d(body, img.shape, False).cpu().numpy().tobytes()) == case["decoded_f64_sha256"]

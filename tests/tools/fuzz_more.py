"""Wider run of the randomised parity sweep of tests/test_gpu_fuzz.py:
python tests/tools/fuzz_more.py 40 600 [slice_lo slice_hi [max slice edge]]"""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from oracle.pyoracle import Oracle
from sperr_amd.api import SperrHip
from test_gpu_fuzz import bits, check_slice, make_case

eng, oracle = SperrHip(), Oracle()
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(lo, hi):
    v, chunks, mode, quality = make_case(1000 + seed)
    want = oracle.comp_3d(v, chunks, mode, quality)
    try:
        got = bytes(eng.compress(torch.from_numpy(v.copy()).cuda(), chunks, quality, mode=mode).cpu().numpy())
        ok = got == want
        dev = torch.from_numpy(np.frombuffer(want, dtype=np.uint8).copy()).cuda()
        for as_float in (True, False):
            ok = ok and np.array_equal(bits(eng.decompress(dev, as_float).cpu().numpy()),
                                       bits(oracle.decomp_3d(want, as_float)))
    except Exception as e:   # noqa: BLE001
        ok = False
        print("exception", e)
    if not ok:
        bad += 1
        print("MISMATCH seed", seed, v.shape, chunks, mode, quality, v.dtype)
print("cases", hi - lo, "bad", bad)

if len(sys.argv) > 4:   # random slices through the 2D entry points (streams cut short included)
    slo, shi = int(sys.argv[3]), int(sys.argv[4])
    maxdim = int(sys.argv[5]) if len(sys.argv) > 5 else 140
    sbad = 0
    for seed in range(slo, shi):
        try:
            check_slice(eng, oracle, seed, maxdim)
        except AssertionError as e:
            sbad += 1
            print("SLICE MISMATCH seed", seed, str(e)[:200])
        except Exception as e:   # noqa: BLE001
            sbad += 1
            print("slice exception seed", seed, e)
    print("slices", shi - slo, "bad", sbad)

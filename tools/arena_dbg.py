import os, sys
sys.path.insert(0, os.getcwd())
import torch
from sperr_amd.api import SperrHip
from sperr_amd.synth import turbulence_torch
eng = SperrHip()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
vol = turbulence_torch((n, n, n), "cuda")
print("== compress", file=sys.stderr, flush=True)
s = eng.compress(vol, (n, n, n), 2.0).clone()
print("== decompress", file=sys.stderr, flush=True)
eng.decompress(s, True)

import os, sys
sys.path.insert(0, os.getcwd())
import torch
from sperr_amd.api import SperrHip
from sperr_amd.synth import turbulence_torch
eng = SperrHip()
vol = turbulence_torch((256, 256, 256), "cuda")
print("== compress", file=sys.stderr, flush=True)
s = eng.compress(vol, (256, 256, 256), 2.0).clone()
print("== decompress", file=sys.stderr, flush=True)
eng.decompress(s, True)

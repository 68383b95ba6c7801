"""Wall time against summed kernel time of one decode / encode of a small batch (8 chunks): how much
of a chunk-farm item is launch overhead."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from sperr_amd.api import SperrHip
from sperr_amd.synth import turbulence_torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
eng = SperrHip()
vol = turbulence_torch((n, n, n), "cuda")
out = torch.empty_like(vol)
cbuf = torch.empty(eng.max_compressed_size(vol.shape, (256,) * 3, 2.0), dtype=torch.uint8, device="cuda")
s = eng.compress(vol, (256, 256, 256), 2.0, out=cbuf)
eng.decompress(s, True, out=out, shape_zyx=vol.shape)
torch.cuda.synchronize()
for name, fn in (("compress", lambda: eng.compress(vol, (256, 256, 256), 2.0, out=cbuf)),
                 ("decompress", lambda: eng.decompress(s, True, out=out, shape_zyx=vol.shape))):
    ts = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    eng.profile(True)
    fn()
    torch.cuda.synchronize()
    eng.profile(False)
    rep = eng.profile_report(with_sum=True)
    ksum = sum(v[2] for v in rep.values())
    nl = sum(v[1] for v in rep.values())
    print(f"{name:10s} {n}^3: wall {min(ts) * 1e3:6.2f} ms   kernels {ksum:6.2f} ms in {nl} launches")
    if len(sys.argv) > 2:
        for k, v in sorted(rep.items(), key=lambda kv: -kv[1][2])[:14]:
            print(f"   {k:28s} {v[2]:7.3f} ms  {v[1]:4d} launches")

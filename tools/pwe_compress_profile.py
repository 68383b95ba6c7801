import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from sperr_amd.api import SperrHip
from sperr_amd.synth import turbulence_torch
eng = SperrHip()
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
vol = turbulence_torch((S, S, S), "cuda")
tol = 0.01027
out_buf = torch.empty(eng.max_compressed_size(vol.shape, (256, 256, 256), 1.0, 3), dtype=torch.uint8, device="cuda")
s = eng.compress(vol, (256, 256, 256), tol, mode=3, out=out_buf)
torch.cuda.synchronize()
ts = []
for _ in range(3):
    t0 = time.perf_counter(); s = eng.compress(vol, (256, 256, 256), tol, mode=3, out=out_buf); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print("PWE compress %d^3: %.1f ms = %.1f GB/s" % (S, min(ts) * 1e3, vol.numel() * 4 / min(ts) / 1e9))
eng.profile(True); eng.compress(vol, (256, 256, 256), tol, mode=3, out=out_buf); torch.cuda.synchronize(); eng.profile(False)
rep = eng.profile_report(with_sum=True)
print("kernel sum %.1f ms in %d launches" % (sum(v[2] for v in rep.values()), sum(v[1] for v in rep.values())))
for k, v in sorted(rep.items(), key=lambda kv: -kv[1][2])[:22]:
    print("   %-30s %8.3f ms %5d launches" % (k, v[2], v[1]))
s = s.clone()
out = torch.empty_like(vol)
eng.decompress(s, True, out=out, shape_zyx=vol.shape); torch.cuda.synchronize()
ts = []
for _ in range(3):
    t0 = time.perf_counter(); eng.decompress(s, True, out=out, shape_zyx=vol.shape); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print("PWE decompress 1024^3: %.1f ms = %.1f GB/s, max err %.4g" % (min(ts) * 1e3, vol.numel() * 4 / min(ts) / 1e9, float((out.double() - vol.double()).abs().max())))
eng.profile(True); eng.decompress(s, True, out=out, shape_zyx=vol.shape); torch.cuda.synchronize(); eng.profile(False)
rep = eng.profile_report(with_sum=True)
for k, v in sorted(rep.items(), key=lambda kv: -kv[1][2])[:10]:
    print("   %-30s %8.3f ms %5d launches" % (k, v[2], v[1]))

import sys, os, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from sperr_amd.api import SperrHip
from oracle.pyoracle import Oracle
eng = SperrHip(); o = Oracle()
v = np.fromfile("tests/golden/wmag17.f32", dtype=np.float32).reshape(17, 17, 17)
want = o.comp_3d(v, (17, 17, 17), 3, 1.0)
dev = torch.from_numpy(np.frombuffer(want, dtype=np.uint8).copy()).cuda()
got = eng.decompress(dev, False).cpu().numpy()
ref = o.decomp_3d(want, False)
# without outliers: truncate the outlier stream off
d = np.abs(got - ref)
print("mismatch count", (d > 0).sum(), "max", d.max())
idx = np.argwhere(d.ravel() > 0).ravel()[:20]
print(idx, got.ravel()[idx], ref.ravel()[idx], (got - ref).ravel()[idx])
print("err vs orig: got", np.abs(got - v).max(), "ref", np.abs(ref - v).max())
import struct
b = bytearray(want)
L = struct.unpack_from('<I', b, 14)[0]
tb = struct.unpack_from('<Q', b, 18 + 18)[0]
sl = 9 + (tb + 7) // 8
cut = bytes(b[:14]) + struct.pack('<I', 17 + sl) + bytes(b[18:18 + 17 + sl])
noout = o.decomp_3d(cut, False)
cr = (ref - noout).ravel(); cg = (got - noout).ravel()
nz = np.argwhere((cr != 0) | (cg != 0)).ravel()
print("ref correctors", (cr != 0).sum(), "gpu correctors", (cg != 0).sum())
bad = [i for i in nz if cr[i] != cg[i]]
for i in bad[:30]:
    print(i, "ref %.3f gpu %.3f" % (cr[i], cg[i]))

import ctypes as C, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from sperr_amd.api import SperrHip
from sperr_amd.synth import turbulence_torch
tol=0.01027; S=1024
eng=SperrHip(); lib=eng.lib
vol=turbulence_torch((S,S,S), torch.device("cuda",0))
hvol=torch.empty(vol.shape,dtype=torch.float32).pin_memory(); hvol.copy_(vol); del vol
libc=C.CDLL(None); libc.free.argtypes=[C.c_void_p]
def comp():
    dst,n=C.c_void_p(None),C.c_size_t(0)
    t0=time.perf_counter()
    rc=lib.sperrhip_comp_3d_farm(hvol.data_ptr(),1,S,S,S,256,256,256,3,tol,0,None,0,C.byref(dst),C.byref(n))
    t1=time.perf_counter(); assert rc==0
    libc.free(dst); return t1-t0
for rep in range(3):
    for w in ("2","3"):
        os.environ["SPERR_HIP_FARM_WORKERS"]=w
        comp()
        ts=[comp() for _ in range(3)]
        print("workers",w,"compress %.1f ms %.1f GB/s"%(min(ts)*1e3, S**3*4/min(ts)/1e9), flush=True)

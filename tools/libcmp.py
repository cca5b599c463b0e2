"""Dump chain states after a few step counts with the library named by KLHR_LIB (compare two builds bit for bit)."""
import os, sys
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT)
import numpy as np
from klhr_amd import _lib
tag=sys.argv[1]; model=sys.argv[2]; D=int(sys.argv[3]); C=int(sys.argv[4])
data=np.array([D-1.0]) if model=="funnel" else np.array([float(D)])
s=_lib.Sampler(_lib.MODEL_IDS[model],data,C,warmup=100)
s.seed(np.arange(C,dtype=np.uint64))
out={}
n=0
for k in (1,1,3,45,50,50):
    s.step(k); n+=k
    out[f"theta{n}"]=s.get("theta").copy(); out[f"ge{n}"]=s.get("grad_evals").copy()
np.savez(os.path.join(ROOT,"gpurun_out",f"libcmp_{tag}.npz"),**out)
print(tag,"done",{k:float(np.abs(v).sum()) for k,v in out.items() if k.startswith("ge")})

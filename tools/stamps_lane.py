"""Cycle split of the lane kernel's iteration (diagnostic build -DKLHR_STAMPS, loaded through KLHR_LIB)."""
import os, sys
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT)
import numpy as np
from klhr_amd import _lib
C=65536
s=_lib.Sampler(_lib.MODEL_IDS["funnel"],np.array([9.0]),C,warmup=100)
s.seed(np.arange(C,dtype=np.uint64)); s.step(150)
s.step(50, record_diag=True)
d=s.get("diag")[::64,:11]
names=["direction","line evals","KL evals (own)","KL evals (shared)","optimiser step","MH","update","draw store","trips","KL trips","shared trips"]
tot=d[:,:8].sum(axis=1).mean()
for k,n in enumerate(names):
    v=d[:,k].mean()
    print(f"{n:20s} {v/50:12.0f} per iteration" + (f"  ({100*v/tot:5.1f} % of cycles)" if k<8 else ""))
print("cycles per iteration", tot/50)

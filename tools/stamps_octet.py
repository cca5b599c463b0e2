"""Cycle split of the octet kernel's iteration (diagnostic build -DKLHR_STAMPS, loaded through KLHR_LIB).
Check the diagnostic build against the production library first (tools/libcmp_iter.py with and without KLHR_LIB):
an earlier octet build with stamps was miscompiled (open issues)."""
import os, sys
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT)
os.environ.setdefault("KLHR_KERNEL","octet")
import numpy as np
from klhr_amd import _lib
model=sys.argv[1] if len(sys.argv)>1 else "funnel"; D=int(sys.argv[2]) if len(sys.argv)>2 else 100; C=int(sys.argv[3]) if len(sys.argv)>3 else 8192
data=np.array([D-1.0]) if model=="funnel" else np.array([float(D)])
if model=="glmm-poisson":
    import json
    d=json.load(open(os.path.join(ROOT,"klhr_amd","stan","glmm-poisson.json"))); data=np.concatenate([[float(d["n"])],d["year"],d["C"]])
s=_lib.Sampler(_lib.MODEL_IDS[model],data,C,warmup=100)
s.seed(np.arange(C,dtype=np.uint64)); s.step(150)
N=20
s.step(N, record_diag=True)
d=s.get("diag")[::8,:8]
names=["direction","evaluations","optimiser step","MH","update","draw store","waiting for the wavefront's other chains","trips"]
tot=d[:,:7].sum(axis=1).mean()
print(f"{model}-{D} x {C} (first chain of every wavefront)")
for k,n in enumerate(names):
    v=d[:,k].mean()
    print(f"{n:42s} {v/N:12.0f} per iteration" + (f"  ({100*v/tot:5.1f} % of cycles)" if k<7 else ""))
print("cycles per iteration", tot/N)

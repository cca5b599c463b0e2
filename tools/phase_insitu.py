"""In-situ phase cost: time 50-iteration launches with one phase left out (KLHR_DEBUG_PHASES), production library."""
import os, sys
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT)
import numpy as np
from klhr_amd import _lib
model=sys.argv[1] if len(sys.argv)>1 else "funnel"; D=int(sys.argv[2]) if len(sys.argv)>2 else 10; C=int(sys.argv[3]) if len(sys.argv)>3 else 65536
data=np.array([D-1.0]) if model=="funnel" else np.array([float(D)])
if model=="glmm-poisson":
    import json
    d=json.load(open(os.path.join(ROOT,"klhr_amd","stan","glmm-poisson.json"))); data=np.concatenate([[float(d["n"])],d["year"],d["C"]])
s=_lib.Sampler(_lib.MODEL_IDS[model],data,C,warmup=100)
s.seed(np.arange(C,dtype=np.uint64)); s.step(150); s.step(1,record_diag=True)
def t(mask,reps=3):
    if mask is None: os.environ.pop("KLHR_DEBUG_PHASES",None)
    else: os.environ["KLHR_DEBUG_PHASES"]=str(mask)
    best=1e9
    for _ in range(reps):
        s.step(50); best=min(best,s.timing()[0])
    return best/50*1000
full=t(None)
print(f"{model}-{D} C={C} full {full:.1f} us/iter")
for name,mask in (("no direction",14),("no fit",13),("no mh",11),("no adapt",7),("fit only",2),("dir+mh+adapt",13)):
    v=t(mask); print(f"  {name:14s} {v:8.1f} us/iter   (delta {full-v:7.1f})", flush=True)
os.environ.pop("KLHR_DEBUG_PHASES",None)

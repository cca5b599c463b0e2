"""Per-chain and per-wavefront fit pass counts of one iteration for any model (octet kernel: 8 chains per wavefront)."""
import os, sys
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT)
import numpy as np
from klhr_amd import _lib
from klhr_amd._lib import DIAG_COLS as DC
col={k:i for i,k in enumerate(DC)}
model=sys.argv[1] if len(sys.argv)>1 else "funnel"; D=int(sys.argv[2]) if len(sys.argv)>2 else 10; C=int(sys.argv[3]) if len(sys.argv)>3 else 65536
per=int(sys.argv[4]) if len(sys.argv)>4 else 8
data=np.array([D-1.0]) if model=="funnel" else np.array([float(D)])
if model=="glmm-poisson":
    import json
    d=json.load(open(os.path.join(ROOT,"klhr_amd","stan","glmm-poisson.json"))); data=np.concatenate([[float(d["n"])],d["year"],d["C"]])
s=_lib.Sampler(_lib.MODEL_IDS[model],data,C,warmup=100)
s.seed(np.arange(C,dtype=np.uint64)); s.step(150)
for rep in range(3):
    s.step(1,record_diag=True); dg=s.get("diag")
    n1=dg[:,col['nfev1']]; n2=dg[:,col['nfev2']]; ps=dg[:,col['passes']]; it2=dg[:,col['nit2']]; fb=dg[:,col['fallbacks']]
    w=ps[:C//per*per].reshape(-1,per).max(1)
    print(f"nfev1 {n1.mean():.1f} nfev2 {n2.mean():.1f} nit2 {it2.mean():.1f} passes/chain {ps.mean():.1f} (max {ps.max():.0f})  per-wavefront max {w.mean():.1f} (p99 {np.percentile(w,99):.0f}, max {w.max():.0f}) fallbacks {fb.mean():.3f}", flush=True)
    s.step(5)

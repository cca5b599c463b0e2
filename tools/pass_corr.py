"""How predictable is a chain's fit cost?  Correlation of per-chain evaluation counts between iterations, and the
per-wavefront maximum if chains were grouped by a running mean of their past counts (octet kernel: 8 per wavefront)."""
import os, sys
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT)
import numpy as np
from klhr_amd import _lib
from klhr_amd._lib import DIAG_COLS as DC
col={k:i for i,k in enumerate(DC)}
model=sys.argv[1] if len(sys.argv)>1 else "funnel"; D=int(sys.argv[2]) if len(sys.argv)>2 else 100; C=int(sys.argv[3]) if len(sys.argv)>3 else 8192
per=int(sys.argv[4]) if len(sys.argv)>4 else 8
data=np.array([D-1.0]) if model=="funnel" else np.array([float(D)])
if model=="glmm-poisson":
    import json
    d=json.load(open(os.path.join(ROOT,"klhr_amd","stan","glmm-poisson.json"))); data=np.concatenate([[float(d["n"])],d["year"],d["C"]])
s=_lib.Sampler(_lib.MODEL_IDS[model],data,C,warmup=100)
s.seed(np.arange(C,dtype=np.uint64)); s.step(200)
rec=[]
for it in range(60):
    s.step(1,record_diag=True); rec.append(s.get("diag")[:,col['passes']].copy())
rec=np.array(rec)
print(f"{model}-{D} x {C}: evals/chain {rec.mean():.1f}")
for lag in (1,5,25,50):
    print(" lag",lag,"corr",np.mean([np.corrcoef(rec[i],rec[i+lag])[0,1] for i in range(0,60-lag,3)]))
past=rec[:10].mean(0); fut=rec[10:]
def wc(order): return np.mean([x[order].reshape(-1,per).max(1).mean() for x in fut])
print(" per-wavefront max: as laid out",wc(np.arange(C))," grouped by mean of 10 past iterations",wc(np.argsort(past))," mean",fut.mean())

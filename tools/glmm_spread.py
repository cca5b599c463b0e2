"""How uneven is the work of the glmm-poisson chains?  Per-chain evaluation counts over a window, per-wavefront (8 chains)
sums of per-iteration maxima (a proxy of the octet kernel's per-wavefront time) and the slowest / mean ratio."""
import os, sys
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT)
import numpy as np, bench
from klhr_amd import _lib
data, D = bench.model_data("glmm-poisson", 45)
C=8192
s=_lib.Sampler(_lib.MODEL_IDS["glmm-poisson"], data, C, warmup=100)
s.seed(np.arange(C,dtype=np.uint64)+1000); s.step(150)
prev=s.get("grad_evals").astype(np.float64)
per=[]
for it in range(40):
    s.step(1); g=s.get("grad_evals").astype(np.float64); per.append(g-prev); prev=g
per=np.array(per)            # (iters, chains)
tot=per.sum(axis=0)
print("evals per iteration per chain: mean %.0f  median %.0f  p99 %.0f  max %.0f" % (per.mean(), np.median(per), np.percentile(per,99), per.max()))
print("per-chain totals over 40 iterations: mean %.0f  max %.0f  (max/mean %.2f); chains above 2x mean: %d" % (tot.mean(), tot.max(), tot.max()/tot.mean(), (tot>2*tot.mean()).sum()))
wm=per.reshape(40, C//8, 8).max(axis=2).sum(axis=0)       # per wavefront: sum over iterations of the slowest chain
print("per-wavefront sum of per-iteration maxima: mean %.0f  max %.0f  (max/mean %.2f);  mean chain total %.0f -> lane utilisation %.2f" % (wm.mean(), wm.max(), wm.max()/wm.mean(), tot.mean(), tot.mean()/wm.mean()))
ac=np.corrcoef(per[:-1].ravel(), per[1:].ravel())[0,1]
print("lag-1 autocorrelation of a chain's evaluation count: %.2f" % ac)

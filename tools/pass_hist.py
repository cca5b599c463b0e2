"""Dump per-chain fit pass counts of one iteration (funnel-10, 65536 chains, after warm-up) for an offline cost model."""
import os, sys
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT)
import numpy as np
from klhr_amd import _lib
from klhr_amd._lib import DIAG_COLS as DC
col={k:i for i,k in enumerate(DC)}
C=65536
s=_lib.Sampler(_lib.MODEL_IDS["funnel"],np.array([9.0]),C,warmup=100)
s.seed(np.arange(C,dtype=np.uint64)); s.step(150)
out=[]
for rep in range(4):
    rho=s.random_direction(); eta=s.fit(rho); dg=s.get("diag")
    out.append(np.stack([dg[:,col['nfev1']],dg[:,col['nfev2']],dg[:,col['passes']]],1).astype(np.int16))
    s.metropolis(eta,rho); s.adapt(); s.step(7)
np.save(os.path.join(ROOT,"gpurun_out","pass_hist.npy"),np.array(out))
print("ok", np.array(out).mean(axis=(0,1)))

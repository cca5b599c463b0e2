"""Phase-by-phase iteration (direction, fit, MH as separate launches) with the library named by KLHR_LIB."""
import os, sys
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT)
import numpy as np
from klhr_amd import _lib
from klhr_amd._lib import DIAG_COLS as DC
col={k:i for i,k in enumerate(DC)}
tag=sys.argv[1]; model=sys.argv[2]; D=int(sys.argv[3]); C=int(sys.argv[4])
data=np.array([D-1.0]) if model=="funnel" else np.array([float(D)])
s=_lib.Sampler(_lib.MODEL_IDS[model],data,C,warmup=100)
s.seed(np.arange(C,dtype=np.uint64))
th0=s.get("theta").copy()
rho=s.random_direction(); eta=s.fit(rho); s.metropolis(eta,rho)
dg=s.get("diag")
np.savez(os.path.join(ROOT,"gpurun_out",f"libcmp3_{tag}.npz"),th0=th0,theta=s.get("theta"),diag=dg,rho=rho,eta=eta,rng=s.get("rng"))
print(tag,"zp",dg[:6,col["zp"]],"acc",dg[:6,col["accept"]])

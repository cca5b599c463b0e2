"""Run each piece of draw() as its own dispatch (for rocprofv3 --pmc: per-dispatch instruction counts)."""
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
s.seed(np.arange(C,dtype=np.uint64))
s.step(50); s.step(50); s.step(20)          # dispatches 1-3 (iter kernels)
rho=s.random_direction()                    # 4: direction
eta=s.fit(rho)                              # 5: fit
v,g=s.kl_eval(eta,rho)                      # 6: one KL eval
o=s.line_eval(eta[:,0],rho)                 # 7: one line eval
s.metropolis(eta,rho)                       # 8: mh
s.adapt()                                   # 9: adapt
s.step(1)                                   # 10: one full iteration
print("ok")

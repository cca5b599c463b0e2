"""Run each piece of draw() as its own dispatch (for rocprofv3 --pmc: per-dispatch instruction counts)."""
import os, sys
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT)
import numpy as np
from klhr_amd import _lib
C=65536
s=_lib.Sampler(_lib.MODEL_IDS["funnel"],np.array([9.0]),C,warmup=100)
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

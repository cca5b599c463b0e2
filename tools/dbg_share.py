"""Chain results must not depend on which other chains share the wavefront (node-shared KL evaluation)."""
import os, sys
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT)
import numpy as np
from klhr_amd import _lib
def run(seeds, n, model="funnel", data=np.array([9.0])):
    s=_lib.Sampler(_lib.MODEL_IDS[model],data,len(seeds),warmup=100)
    s.seed(np.asarray(seeds,dtype=np.uint64))
    s.step(n)
    return s.get("theta"), s.get("rng")
for model,data in (("funnel",np.array([9.0])),("normal",np.array([2.0]))):
    ref_t,ref_r=run(np.arange(64),160,model,data)
    for sub in (np.arange(8), np.arange(16), np.arange(3,30), np.array([5]), np.arange(40)):
        t,r=run(sub,160,model,data)
        print(model, "C", len(sub), "theta bitwise equal:", np.array_equal(t, ref_t[sub]), "rng equal:", np.array_equal(r, ref_r[sub]), "max diff", np.abs(t-ref_t[sub]).max(), flush=True)

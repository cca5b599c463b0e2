"""One recorded iteration with the library named by KLHR_LIB: dump diag / rho / eta / theta."""
import os, sys
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT)
import numpy as np
from klhr_amd import _lib
tag=sys.argv[1]; model=sys.argv[2]; D=int(sys.argv[3]); C=int(sys.argv[4])
data=np.array([D-1.0]) if model=="funnel" else np.array([float(D)])
s=_lib.Sampler(_lib.MODEL_IDS[model],data,C,warmup=100)
s.seed(np.arange(C,dtype=np.uint64))
th0=s.get("theta").copy()
s.step(1,record_diag=True)
np.savez(os.path.join(ROOT,"gpurun_out",f"libcmp2_{tag}.npz"),th0=th0,theta=s.get("theta"),diag=s.get("diag"),rho=s.get("rho"),eta=s.get("eta"),rng=s.get("rng"))
print(tag,"ok")

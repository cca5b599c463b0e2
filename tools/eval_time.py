import os, sys, numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT)
from klhr_amd import _lib
C=65536
s=_lib.Sampler(_lib.MODEL_IDS["funnel"],np.array([9.0]),C,warmup=100)
s.seed(np.arange(C,dtype=np.uint64)); s.step(120)
rho=s.random_direction(); eta=s.fit(rho)
res={}
for R in (1,101,201):
    os.environ["KLHR_PROBE_REPEAT"]=str(R)
    s.kl_eval(eta,rho); a=s.timing()[0]
    s.line_eval(eta[:,0],rho); b=s.timing()[0]
    res[R]=(a,b)
    print(R, "KL ms", a, "line ms", b, flush=True)
kl=(res[201][0]-res[101][0])/100*1e3; ln=(res[201][1]-res[101][1])/100*1e3
print(f"per KL eval {kl:.2f} us  per line eval {ln:.2f} us  (all 65536 chains; 1 wave/SIMD) -> cycles@2.1GHz: {kl*2100:.0f} / {ln*2100:.0f}")

import sys, os, numpy as np
sys.path.insert(0,'/root/repo')
from klhr_amd import BSModel, KLHRSINH
STAN='/root/repo/klhr_amd/stan'
m=BSModel(STAN+'/funnel.stan',STAN+'/funnel.json')
algo=KLHRSINH(m,seed=11,warmup=100,overrelaxed=False,chains=1024)
import time
for k in range(8):
    t=time.time(); dr=algo.draws(500); dt=time.time()-t
    v=dr[:,:,0]; print(k, "v mean %.3f sd %.3f  q01 %.2f q99 %.2f"%(v.mean(), v.std(), np.quantile(v,0.01), np.quantile(v,0.99)), "alpha1 sd %.3f"%dr[:,:,1].std(), "time %.2f"%dt, flush=True)

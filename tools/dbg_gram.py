import sys, numpy as np
sys.path.insert(0,'/root/repo')
from klhr_amd import _lib
np.set_printoptions(linewidth=250, precision=3, suppress=True)
s=_lib.Sampler(_lib.MODEL_IDS["funnel"],np.array([9.0]),8,warmup=100)
s.seed(np.arange(8,dtype=np.uint64)+5); s.step(1,record_draws=True)
X=s.draws(0,1).reshape(-1,10); g=s.pool_gram(0,1)
G=g[11:].reshape(10,10); R=X.T@X
print("count",g[0],"sums ok",np.allclose(g[1:11],X.sum(0)))
print((np.abs(G-R)<1e-9).astype(int))
print(G[:4,:6]); print(R[:4,:6])

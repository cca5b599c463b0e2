"""Phase breakdown with the PRODUCTION library: time each piece of draw() as its own launch (HIP events)."""
import os, sys, json, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT)
import numpy as np
from klhr_amd import _lib
model=sys.argv[1] if len(sys.argv)>1 else "funnel"; D=int(sys.argv[2]) if len(sys.argv)>2 else 10; C=int(sys.argv[3]) if len(sys.argv)>3 else 65536
if model=="funnel": data=np.array([D-1.0])
elif model=="glmm-poisson":
    d=json.load(open(os.path.join(ROOT,"klhr_amd","stan","glmm-poisson.json"))); data=np.concatenate([[float(d["n"])],d["year"],d["C"]]); D=45
else: data=np.array([float(D)])
def log(*a):
    print(*a, flush=True)
s=_lib.Sampler(_lib.MODEL_IDS[model],data,C,warmup=100)
s.seed(np.arange(C,dtype=np.uint64)); log("seeded")
s.step(120); log("warm")
s.step(50); ms,_=s.timing(); log(f"{model}-{D} C={C}: full step 50 iters {ms:.2f} ms -> {ms/50*1000:.1f} us/iter, {C*50/ms*1e3:.3e} it/s")
res={}
for rep in range(3):
    rho=s.random_direction(); res.setdefault("direction",[]).append(s.timing()[0])
    eta=s.fit(rho); res.setdefault("fit",[]).append(s.timing()[0])
    dg=s.get("diag")
    v,g=s.kl_eval(eta,rho); res.setdefault("one KL pass",[]).append(s.timing()[0])
    o=s.line_eval(eta[:,0],rho); res.setdefault("one line pass",[]).append(s.timing()[0])
    s.metropolis(eta,rho); res.setdefault("mh",[]).append(s.timing()[0])
    s.adapt(); res.setdefault("adapt",[]).append(s.timing()[0])
from klhr_amd._lib import DIAG_COLS as DC
col={k:i for i,k in enumerate(DC)}
log("passes/iter (per chain) mean", dg[:,col['passes']].mean(), "max-of-8 mean", dg[:,col['passes']].reshape(-1,8).max(axis=1).mean(), "nfev1", dg[:,col['nfev1']].mean(), "nfev2", dg[:,col['nfev2']].mean())
for k,v in res.items(): log(f"  {k:14s} {min(v)*1000:9.1f} us")

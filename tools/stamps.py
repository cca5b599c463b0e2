"""Diagnostic: build the library with -DKLHR_STAMPS and print the share of wave cycles per phase (NOT a benchmark)."""
import os, subprocess, sys
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out=os.path.join(ROOT,"klhr_amd","lib","libklhr_hip_stamps.so")
if "--build" in sys.argv:
    subprocess.check_call(["/opt/rocm/bin/hipcc","--offload-arch=gfx950","-O3","-std=c++17","-fPIC","-shared","-fvisibility=hidden","-DKLHR_STAMPS","-o",out,os.path.join(ROOT,"klhr_amd","csrc","klhr_api.hip")])
    sys.exit(0)
os.environ["KLHR_LIB"]=out
sys.path.insert(0,ROOT)
import numpy as np
from klhr_amd import _lib
model=sys.argv[1] if len(sys.argv)>1 else "funnel"; D=int(sys.argv[2]) if len(sys.argv)>2 else 10; C=int(sys.argv[3]) if len(sys.argv)>3 else 65536
import json
if model=="funnel": data=np.array([D-1.0])
elif model=="glmm-poisson":
    d=json.load(open(os.path.join(ROOT,"klhr_amd","stan","glmm-poisson.json"))); data=np.concatenate([[float(d["n"])],d["year"],d["C"]])
else: data=np.array([float(D)])
s=_lib.Sampler(_lib.MODEL_IDS[model],data,C,warmup=100)
s.seed(np.arange(C,dtype=np.uint64))
s.step(100)
s.step(50,record_draws=True,record_diag=True)
ms,nl=s.timing()
d=s.get("diag")[::8,:8]
tot=d[:,:7].sum(axis=1)
names=["direction","eval","advance","mh","adapt","out","idle(waiting for other octets)"]
print(f"{model}-{D} C={C}: launch {ms:.2f} ms for 50 iters; cycles/wave/iter {tot.mean()/50:.0f}; passes/iter {d[:,7].mean()/50:.1f}")
for i,n in enumerate(names): print(f"  {n:34s} {100*d[:,i].sum()/tot.sum():5.1f} %   {d[:,i].mean()/50:9.0f} cyc/iter")

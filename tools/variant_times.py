"""Throughput of the sampler variants / flags on the headline workload (funnel-10, 65536 chains)."""
import os, sys
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT)
import numpy as np
from klhr_amd import _lib
C=65536
for name,kw in (("klhr_sinh",{}),("klhr_sinh overrelaxed",dict(overrelaxed=True)),("klhr (gauss)",dict(variant=1)),("klhr overrelaxed",dict(variant=1,overrelaxed=True)),
                ("sub_klhr_sinh",dict(variant=2)),("sub overrelaxed",dict(variant=2,overrelaxed=True)),("slice (octet kernel)",dict(variant=3,tol=1e-12,eigen_method_one=True)),("eigen_method_one",dict(eigen_method_one=True)),("scale_dir_cov",dict(scale_dir_cov=True))):
    s=_lib.Sampler(_lib.MODEL_IDS["funnel"],np.array([9.0]),C,warmup=100,**kw)
    s.seed(np.arange(C,dtype=np.uint64)); s.step(150)
    best=1e9
    for _ in range(3):
        s.step(50); best=min(best,s.timing()[0])
    print(f"{name:24s} {best/50*1000:8.1f} us/iter  {C*50/best*1e3:.3e} it/s  grad_evals/iter {s.get('grad_evals').mean()/ (150+150):.1f}", flush=True)
    del s

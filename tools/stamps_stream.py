"""Cycle split of the split pipeline's streaming kernel (diagnostic build -DKLHR_STAMPS loaded through KLHR_LIB):
KLHR_BUILD_OUT=klhr_amd/lib/libklhr_stamps.so KLHR_EXTRA_FLAGS=-DKLHR_STAMPS python -m klhr_amd.build --force
KLHR_LIB=klhr_amd/lib/libklhr_stamps.so python tools/stamps_stream.py ill-normal 100 16384"""
import os, sys
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT)
os.environ.setdefault("KLHR_KERNEL","split")
import numpy as np
from klhr_amd import _lib
model=sys.argv[1] if len(sys.argv)>1 else "funnel"; D=int(sys.argv[2]) if len(sys.argv)>2 else 100; C=int(sys.argv[3]) if len(sys.argv)>3 else 8192
data=np.array([D-1.0]) if model=="funnel" else np.array([float(D)])
s=_lib.Sampler(_lib.MODEL_IDS[model],data,C,warmup=100)
s.seed(np.arange(C,dtype=np.uint64)); s.step(150)
s.set("diag", np.zeros((C,_lib.NDIAG)))
N=20
s.step(N)
d=s.get("diag")
tw=16 if "tile" in s.kernel_name() or "stream16" in s.kernel_name() else 64
print(s.kernel_name())
heads=d[1::tw]; heads=heads[heads[:,7]>0]           # pair mode: the head's wavefront reports in the tile's second row
rows=d[0::tw]; rows=rows[rows[:,7]>0]
names=["load state","MH scalars","T1 (theta', moments, u)","CCIPCA / closure","normals","direction + line sums + fit normals","store state"]
tot=rows[:,:7].sum(axis=1).mean()
print(f"{model}-{D} x {C}: {len(rows)} tiles, {rows[:,7].mean():.0f} launches each")
for k,n in enumerate(names):
    v=rows[:,k].mean()
    print(f"{n:40s} {v/N:12.0f} cycles per iteration  ({100*v/tot:5.1f} %)")
print("cycles per iteration", tot/N)
if len(heads):
    print("pair mode, the head's wavefront:")
    for k,n in enumerate(names):
        print(f"{n:40s} {heads[:,k].mean()/N:12.0f} cycles per iteration")
if rows[:,8].any():
    print(f"tile kernel: fit {rows[:,8].mean()/N:.0f} cycles per iteration, wait for the tile's other wavefront after it {rows[:,9].mean()/N:.0f}")
    if len(heads): print(f"             second wavefront: fit {heads[:,8].mean()/N:.0f}, wait {heads[:,9].mean()/N:.0f}")

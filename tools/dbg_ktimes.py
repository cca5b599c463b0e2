"""Serialised per-kernel times of one iteration of the split pipeline (kernel profiling mode) beside the overlapped wall time:
python tools/dbg_ktimes.py model D C [groups]"""
import sys, os, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT)
if len(sys.argv) > 4: os.environ["KLHR_GROUPS"] = sys.argv[4]
import numpy as np
from klhr_amd import _lib
m, D, C = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
data = np.array([D-1.0]) if m == "funnel" else np.array([float(D)])
s = _lib.Sampler(_lib.MODEL_IDS[m], data, C, warmup=100)
s.seed(np.arange(C, dtype=np.uint64)); s.step(150); s.sync()
t0 = time.perf_counter(); s.step(50); s.sync(); t1 = time.perf_counter()
print(f"{m}-{D} x {C}: {s.kernel_name()}  wall {1e6*(t1-t0)/50:.1f} us / iteration = {C*50/(t1-t0):.3e} it/s")
if os.environ.get("DBG_WALL_ONLY"): sys.exit(0)
s.kernel_profiling(True); s.step(20)
names = s.kernel_names(); ms, n = s.kernel_times()
for k, nm in enumerate(names): print(f"   {nm:20s} mean {1e3*ms[k]:9.1f} us x {n[k]} launches = {1e3*ms[k]*n[k]/20:9.1f} us / iteration (serialised)")
print(f"   sum {1e3*sum(ms[k]*n[k] for k in range(len(names)))/20:9.1f} us / iteration")

"""Cycle split of the free-running octet kernel (diagnostic build -DKLHR_STAMPS, loaded through KLHR_LIB; compare the build with
production first: tools/libcmp_iter.py).   python tools/stamps_free.py glmm-poisson 45 8192"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ.setdefault("KLHR_KERNEL", "free")
import numpy as np
from klhr_amd import _lib
model = sys.argv[1] if len(sys.argv) > 1 else "glmm-poisson"; D = int(sys.argv[2]) if len(sys.argv) > 2 else 45; C = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
data = np.array([D - 1.0]) if model == "funnel" else np.array([float(D)])
if model == "glmm-poisson":
    import json
    d = json.load(open(os.path.join(ROOT, "klhr_amd", "stan", "glmm-poisson.json"))); data = np.concatenate([[float(d["n"])], d["year"], d["C"]])
s = _lib.Sampler(_lib.MODEL_IDS[model], data, C, warmup=100)
s.seed(np.arange(C, dtype=np.uint64)); s.step(150)
N = 50
s.step(N, record_diag=True)
assert "free" in s.kernel_name(), s.kernel_name()
d = s.get("diag")[::8, :8]
names = ["service (tail + head, whole wavefront)", "site L (line objective, MH values)", "site K (KL objective)", "advance (optimiser, transitions)"]
tot = d[:, :4].sum(axis=1).mean()
print(f"{model}-{D} x {C}, {N} iterations in one launch (per wavefront = 8 chains)")
for k, n in enumerate(names):
    v = d[:, k].mean()
    print(f"{n:42s} {v/N:12.0f} cycles per wavefront-iteration  ({100*v/tot:5.1f} %)")
print(f"{'trips':42s} {d[:,4].mean()/N:12.1f} per wavefront-iteration; with a line evaluation {d[:,5].mean()/N:.1f}, with a KL evaluation {d[:,6].mean()/N:.1f}; "
      f"trips in which octet 0 had work {d[:,7].mean()/N:.1f}")
print("cycles per wavefront-iteration", tot / N, " slowest / mean wavefront", d[:, :4].sum(axis=1).max() / tot)

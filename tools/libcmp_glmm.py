"""theta / rng / grad_evals of glmm-poisson-45 after N draws, saved to a file (arg 1) — run once per library (KLHR_LIB) and compare:
python tools/libcmp_glmm.py out.npz [C] [N]; python tools/libcmp_glmm.py --cmp a.npz b.npz"""
import sys, os, json
import numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT)
if sys.argv[1] == "--cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    for k in a.files:
        same = np.array_equal(a[k], b[k], equal_nan=True)
        print(k, "identical" if same else "DIFFERENT max|d| %.3e in %d of %d rows" % (np.nanmax(np.abs(a[k].astype(float) - b[k].astype(float))), (a[k] != b[k]).reshape(len(a[k]), -1).any(axis=1).sum(), len(a[k])))
    sys.exit(0)
from klhr_amd import _lib
C = int(sys.argv[2]) if len(sys.argv) > 2 else 512; N = int(sys.argv[3]) if len(sys.argv) > 3 else 60
d=json.load(open(os.path.join(ROOT,"klhr_amd","stan","glmm-poisson.json"))); data=np.concatenate([[float(d["n"])],d["year"],d["C"]])
out = {}
for variant in (0, 1, 2):
    for over in (False, True):
        s=_lib.Sampler(_lib.MODEL_IDS["glmm-poisson"],data,C,warmup=100,variant=variant,overrelaxed=over)
        s.seed(np.arange(C,dtype=np.uint64)+5); s.step(N)
        for f in ("theta","rng","grad_evals","acc"):
            out[f"{f}_v{variant}_o{int(over)}"] = s.get(f)
np.savez(sys.argv[1], **out)
print("saved", sys.argv[1], os.environ.get("KLHR_LIB", "production library"))

"""Node-parallel fit (KLHR_FITN=1) against the lane = chain fit kernels (KLHR_FITN=0): chain state after N draws must be bit-identical.
python tools/dbg_fitn.py out.npz [model D C N]  (run once per setting), python tools/dbg_fitn.py --cmp a.npz b.npz"""
import sys, os
import numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT)
if sys.argv[1] == "--cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3]); bad = 0
    for k in a.files:
        same = np.array_equal(a[k], b[k], equal_nan=True); bad += not same
        print(k, "identical" if same else "DIFFERENT in %d of %d rows" % ((a[k] != b[k]).reshape(len(a[k]), -1).any(axis=1).sum(), len(a[k])))
    sys.exit(1 if bad else 0)
os.environ.setdefault("KLHR_KERNEL", "split")
from klhr_amd import _lib
m = sys.argv[2] if len(sys.argv) > 2 else "funnel"; D = int(sys.argv[3]) if len(sys.argv) > 3 else 100
C = int(sys.argv[4]) if len(sys.argv) > 4 else 500; N = int(sys.argv[5]) if len(sys.argv) > 5 else 120
data = np.array([D-1.0]) if m == "funnel" else np.array([float(D)])
out = {}
for variant in (0, 1, 2):
    for over in (False, True):
        s=_lib.Sampler(_lib.MODEL_IDS[m],data,C,warmup=100,variant=variant,overrelaxed=over)
        s.seed(np.arange(C,dtype=np.uint64)+5); s.step(N, record_diag=True)
        for f in ("theta","rng","grad_evals","acc","cov","eigvecs","diag","status"):
            out[f"{f}_v{variant}_o{int(over)}"] = s.get(f)
        name = s.kernel_name()
np.savez(sys.argv[1], **out)
print("saved", sys.argv[1], name, "KLHR_FITN =", os.environ.get("KLHR_FITN"))

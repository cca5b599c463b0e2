"""Find the first draw at which the node-parallel fit (KLHR_FITN=1) and the lane = chain fit kernels (KLHR_FITN=0) leave different
bits, teacher-forced draw by draw (both start every draw from the lane form's state): python tools/dbg_fitn_trace.py model D C n"""
import sys, os
import numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT)
os.environ["KLHR_FUSED"] = "0"
from klhr_amd import _lib
m, D, C, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
data = np.array([D-1.0]) if m == "funnel" else np.array([float(D)])
col = {k: i for i, k in enumerate(_lib.DIAG_COLS)}
def mk(fitn):
    os.environ["KLHR_FITN"] = fitn
    s = _lib.Sampler(_lib.MODEL_IDS[m], data, C, warmup=100, **({'scale_clip': float(os.environ['DBG_CLIP'])} if os.environ.get('DBG_CLIP') else {}))
    s.seed(np.arange(C, dtype=np.uint64) + 1000)
    return s
a, b = mk("0"), mk("1")
print(a.kernel_name(), "|", b.kernel_name())
for it in range(n):
    a.step(1, record_diag=True); b.step(1, record_diag=True)
    ea, eb = a.get("eta"), b.get("eta")
    da, db = a.get("diag"), b.get("diag")
    bad = np.flatnonzero((ea != eb).any(axis=1) | (a.get("theta") != b.get("theta")).any(axis=1))
    if len(bad):
        for c in bad[:4]:
            print(f"draw {it} chain {c}: eta lane {ea[c]} node {eb[c]}")
            for k in ("mode", "init0", "init1", "init2", "init3", "nfev1", "nfev2", "nit2", "status2", "fun2", "fallbacks", "passes", "zp", "accept"):
                print(f"     {k:10s} {da[c, col[k]]!r:26} {db[c, col[k]]!r}")
        break
else:
    print("no difference in", n, "draws")

"""Distribution of the fit's passes per chain-iteration (objective evaluations: nfev1 mode search, nfev2 KL fit) and how much of a
launch the slowest chain of a wavefront / of the launch adds: python tools/pass_dist.py model D C"""
import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT)
import numpy as np
from klhr_amd import _lib
m, D, C = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
data = np.array([D-1.0]) if m == "funnel" else np.array([float(D)])
s = _lib.Sampler(_lib.MODEL_IDS[m], data, C, warmup=100)
s.seed(np.arange(C, dtype=np.uint64)); s.step(150)
col = {k: i for i, k in enumerate(_lib.DIAG_COLS)}
for it in range(3):
    s.step(1, record_diag=True)
    d = s.get("diag")
    n1, n2 = d[:, col["nfev1"]], d[:, col["nfev2"]]
    q = lambda x: " ".join("%g" % np.percentile(x, p) for p in (50, 90, 99, 99.9, 100))
    w2 = n2[: C // 64 * 64].reshape(-1, 64)
    print(f"draw {it}: nfev1 p50/90/99/99.9/max {q(n1)} | nfev2 {q(n2)} | mean {n2.mean():.1f}; per 64-chain wavefront: mean of max {w2.max(axis=1).mean():.1f}, "
          f"lanes busy {n2.mean() / w2.max(axis=1).mean():.2f}; launch max / mean-of-wavefront-max {n2.max() / w2.max(axis=1).mean():.2f}")

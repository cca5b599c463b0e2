"""Fused tile kernel (KLHR_FUSED=1) against the kernel sequence (KLHR_FUSED=0): python tools/dbg_fused.py model D C n [record]"""
import sys, os, subprocess, json
import numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT)
if len(sys.argv) > 6:       # worker
    os.environ["KLHR_FUSED"] = sys.argv[6]
    from klhr_amd import _lib
    m, D, C, n, rec = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    data = np.array([D-1.0]) if m == "funnel" else np.array([float(D)])
    s = _lib.Sampler(_lib.MODEL_IDS[m], data, C, warmup=100)
    s.seed(np.arange(C, dtype=np.uint64) + 1000)
    out = {}
    done = 0
    for k in range(n):
        s.step(1, record_draws=bool(rec)) if os.environ.get("DBG_ONE") else None
    if not os.environ.get("DBG_ONE"): s.step(n, record_draws=bool(rec))
    for f in ("theta", "rng", "cov", "eigvecs", "pca_v", "acc", "grad_evals", "counters", "om_m", "eta"):
        out[f] = s.get(f)
    np.savez(f"/tmp/fused_{sys.argv[6]}.npz", **out)
    print(s.kernel_name())
    sys.exit(0)
for mode in ("0", "1"):
    subprocess.check_call([sys.executable, __file__] + sys.argv[1:6] + [mode])
a, b = np.load("/tmp/fused_0.npz"), np.load("/tmp/fused_1.npz")
for k in a.files:
    same = np.array_equal(a[k], b[k], equal_nan=True)
    if same: print(k, "identical")
    else:
        rows = np.flatnonzero((a[k] != b[k]).reshape(len(a[k]), -1).any(axis=1))
        print(k, "DIFFERENT rows", rows[:10], "max|d|", np.nanmax(np.abs(a[k].astype(float) - b[k].astype(float))))

"""First 64 chains of a big launch against a 64-chain launch under different mapping knobs: python tools/dbg_sizes.py model D Cbig n"""
import sys, os, subprocess
import numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT)
if len(sys.argv) > 6:
    for kv in sys.argv[7:]:
        k, v = kv.split("="); os.environ[k] = v
    from klhr_amd import _lib
    m, D, C, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[5]), int(sys.argv[4])
    data = np.array([D-1.0]) if m == "funnel" else np.array([float(D)])
    s = _lib.Sampler(_lib.MODEL_IDS[m], data, C, warmup=100)
    s.seed(np.arange(C, dtype=np.uint64) + 1000); s.step(n)
    np.savez(sys.argv[6], **{f: s.get(f)[:64] for f in ("theta", "rng", "cov", "eigvecs", "pca_v", "acc", "grad_evals", "counters", "om_m", "eta")})
    print(sys.argv[6], s.kernel_name(), sys.argv[7:]); sys.exit(0)
m, D, Cbig, n = sys.argv[1:5]
subprocess.check_call([sys.executable, __file__, m, D, Cbig, n, Cbig, "/tmp/sz_big.npz"])
a = np.load("/tmp/sz_big.npz")
for knobs in ([], ["KLHR_FUSED=0"], ["KLHR_FUSED=0", "KLHR_STREAM_PAIR=0"], ["KLHR_FUSED=0", "KLHR_FITN=0"], ["KLHR_FUSED=0", "KLHR_TILE=64", "KLHR_FITN=0"]):
    subprocess.check_call([sys.executable, __file__, m, D, Cbig, n, "64", "/tmp/sz_small.npz"] + knobs)
    b = np.load("/tmp/sz_small.npz")
    bad = [k for k in a.files if not np.array_equal(a[k], b[k], equal_nan=True)]
    print("   vs big:", "identical" if not bad else "DIFFERENT " + ",".join(bad) + " rows " + str(np.flatnonzero((a["theta"] != b["theta"]).any(axis=1))[:8]))

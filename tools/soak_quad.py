"""Long soak of quad mode against lane = chain: the same chains for many draws through the tile kernel (16-chain tiles, four lanes per
chain, quad generator) and through the 64-chain kernel sequence; generator state, theta, adaptation state and status must be the same
bits at the end.  10^10 normals per run: every branch of the ziggurat (tail, wedge, repeated rejections) goes through the quad generator's
park / re-synchronise path thousands of times.   python tools/soak_quad.py [stem D chains draws]"""
import os, sys, subprocess, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
if len(sys.argv) > 5:
    stem, D, C, n, tag = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    from klhr_amd import _lib
    kw = dict(eigen_method_one=True) if tag.endswith("e1") else {}
    s = _lib.Sampler(_lib.MODEL_IDS[stem], np.array([float(D - 1 if stem == "funnel" else D)]), C, warmup=1000, **kw)
    s.seed(np.arange(C, dtype=np.uint64) + 77)
    t0 = time.time(); done = 0
    while done < n:
        s.step(min(5000, n - done)); done += min(5000, n - done)
    s.sync()
    print(tag, s.kernel_name(), f"{time.time() - t0:.1f} s", "status", np.unique(s.get("status")), flush=True)
    np.savez(f"/tmp/soakq_{tag}.npz", **{f: s.get(f) for f in ("rng", "theta", "mean", "cov", "eigvecs", "eigvals", "acc", "grad_evals", "counters", "status")})
else:
    cfgs = [("ill-normal", 100, 4096, 30000, ""), ("funnel", 100, 2048, 20000, ""), ("normal", 37, 1024, 40000, "e1")] if len(sys.argv) < 5 else [(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), "")]
    bad = 0
    for stem, D, C, n, fl in cfgs:
        for tag, env in (("q" + fl, {"KLHR_KERNEL": "split", "KLHR_TILE": "16"}), ("l" + fl, {"KLHR_KERNEL": "split", "KLHR_TILE": "64", "KLHR_FITN": "1"})):
            subprocess.run([sys.executable, __file__, stem, str(D), str(C), str(n), tag], env=dict(os.environ, **env), check=True)
        a, b = np.load(f"/tmp/soakq_q{fl}.npz"), np.load(f"/tmp/soakq_l{fl}.npz")
        for f in a.files:
            same = np.array_equal(a[f], b[f], equal_nan=True)
            bad += not same
            print(f"  {stem}-{D} x {C} x {n} draws: {f:10s} {'same bits' if same else 'DIFFERENT'}")
    print("soak_quad done, differing fields:", bad)
    sys.exit(1 if bad else 0)

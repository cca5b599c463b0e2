"""For separable targets of dimension <= 26: the lane kernel against the split pipeline (tile kernel, kernel sequence), the same chains for n draws
through the window closures — draws, generator, adaptation state, evaluation counts must be the same bits.   python tools/lane_vs_split.py"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
FIELDS = ("rng", "theta", "mean", "cov", "eigvecs", "eigvals", "om_m", "om_v", "pca_v", "acc", "grad_evals", "counters", "status")
if len(sys.argv) > 1:
    stem, D, C, n, tag, opts = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], eval(sys.argv[6])
    from klhr_amd import _lib
    s = _lib.Sampler(_lib.MODEL_IDS[stem], np.array([float(D - 1 if stem == "funnel" else D)]), C, warmup=150, **opts)
    s.seed(np.arange(C, dtype=np.uint64) + 900)
    s.step(n, record_draws=True)
    print("  ", tag, s.kernel_name(), flush=True)
    np.savez(f"/tmp/lvs_{tag}.npz", draws=s.draws(0, n), **{f: s.get(f) for f in FIELDS})
else:
    bad = 0
    for stem, D, opts in (("funnel", 10, {}), ("funnel", 10, dict(variant=1)), ("funnel", 10, dict(variant=2)), ("funnel", 10, dict(overrelaxed=True)),
                          ("funnel", 10, dict(eigen_method_one=True)), ("funnel", 10, dict(scale_dir_cov=True)), ("ill-normal", 26, {}), ("normal", 3, {}), ("funnel", 2, {})):
        print(stem, D, opts)
        envs = {"lane": {"KLHR_KERNEL": "lane"}, "tile": {"KLHR_KERNEL": "split", "KLHR_FUSED": "1"}, "seq": {"KLHR_KERNEL": "split", "KLHR_FUSED": "0"},
                "seq64": {"KLHR_KERNEL": "split", "KLHR_FUSED": "0", "KLHR_TILE": "64", "KLHR_FITN": "0"}}
        for tag, env in envs.items():
            subprocess.run([sys.executable, __file__, stem, str(D), "200", "260", tag, repr(opts)], env=dict(os.environ, **env), check=True)
        a = np.load("/tmp/lvs_lane.npz")
        for tag in ("tile", "seq", "seq64"):
            b = np.load(f"/tmp/lvs_{tag}.npz")
            diff = [f for f in a.files if not np.array_equal(a[f], b[f], equal_nan=True)]
            first = -1
            if "draws" in diff:
                d = (a["draws"] != b["draws"]).any(axis=(1, 2)); first = int(np.argmax(d))
            bad += bool(diff)
            print(f"   lane vs {tag}: {'same bits' if not diff else 'DIFFERENT: ' + ', '.join(diff) + f' (first differing draw {first})'}")
    print("lane_vs_split done, differing comparisons:", bad)

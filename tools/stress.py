"""Stress run: odd chain counts (partially filled wavefronts), every model x variant, both kernel mappings.  Each case
in a child process under a timeout, so that a kernel that failed to terminate would be reported, not waited for."""
import os, sys, subprocess, json, itertools
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,"tests"))
if len(sys.argv) > 1 and sys.argv[1] == "one":
    import numpy as np
    from conftest import model_spec
    from parity_util import make_sampler
    name, variant, C, n, over = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
    stem, data = model_spec(name)
    kw = dict(warmup=100, variant=variant, overrelaxed=bool(over))
    if variant == 3: kw.update(tol=1e-12, eigen_method_one=True, overrelaxed=False)
    s = make_sampler(stem, data, C, **kw)
    s.seed(np.arange(C, dtype=np.uint64) + 17)
    s.step(n)
    th, st = s.get("theta"), s.get("status")
    print(json.dumps({"finite": bool(np.isfinite(th).all()), "status_max": int(st.max()), "acc": float(s.get("acc").mean()), "kernel": s.kernel_name()}))
    sys.exit(0)
models = ["funnel10", "funnel100", "illnormal100", "glmm45", "normal2", "earnings", "ar1x100"]
bad = 0
for name, variant, C, over in itertools.product(models, (0, 1, 2, 3), (1, 7, 9, 63, 65), (0, 1)):
    if variant == 3 and over: continue
    n = 60 if name in ("glmm45",) else 120
    try:
        r = subprocess.run([sys.executable, __file__, "one", name, str(variant), str(C), str(n), str(over)], capture_output=True, text=True, timeout=90)
        ok = r.returncode == 0 and '"finite": true' in r.stdout and '"status_max": 0' in r.stdout
        if not ok:
            bad += 1
            print("FAIL", name, variant, C, over, r.stdout.strip()[-200:], r.stderr.strip()[-300:], flush=True)
    except subprocess.TimeoutExpired:
        bad += 1
        print("TIMEOUT", name, variant, C, over, flush=True)
print("stress done, failures:", bad, flush=True)

"""Extreme chain states must neither spin nor fault: NaN / inf / huge theta, funnel necks and mouths."""
import os, sys, subprocess
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,"tests"))
import numpy as np
if len(sys.argv) > 1:
    from conftest import model_spec
    from parity_util import make_sampler
    name, case, variant = sys.argv[1], sys.argv[2], int(sys.argv[3])
    stem, data = model_spec(name)
    kw = dict(warmup=100, variant=variant)
    if variant == 3: kw["tol"] = 1e-12
    s = make_sampler(stem, data, 16, **kw); s.seed(np.arange(16, dtype=np.uint64))
    th = s.get("theta")
    if case == "nan": th[:] = np.nan
    elif case == "inf": th[:, 0] = np.inf
    elif case == "ninf": th[:, 0] = -np.inf
    elif case == "huge": th[:] = 1e300
    elif case == "neck": th[:, 0] = -30.0
    elif case == "mouth": th[:, 0] = 30.0
    elif case == "neck700": th[:, 0] = -700.0
    elif case == "mixed": th[::2] = np.nan
    s.set("theta", th)
    s.step(30)
    t2 = s.get("theta"); st = s.get("status")
    print(case, "finite chains", int(np.isfinite(t2).all(axis=1).sum()), "of 16; status values", sorted(set(st.tolist())))
    sys.exit(0)
bad = 0
for name in ("funnel10", "illnormal100", "glmm45"):
    for variant in (0, 1, 2, 3):
        for case in ("nan", "inf", "ninf", "huge", "neck", "mouth", "neck700", "mixed"):
            try:
                r = subprocess.run([sys.executable, __file__, name, case, str(variant)], capture_output=True, text=True, timeout=60)
                if r.returncode != 0:
                    bad += 1; print("FAIL", name, variant, case, r.stderr.strip()[-200:], flush=True)
                else:
                    print(name, "v%d" % variant, r.stdout.strip(), flush=True)
            except subprocess.TimeoutExpired:
                bad += 1; print("TIMEOUT", name, variant, case, flush=True)
print("extreme-state run done, failures:", bad)

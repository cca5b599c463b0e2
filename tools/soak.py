"""Soak run of the BASELINE configurations: thousands of iterations at full chain counts; all states finite, status clear,
and the pooled post-warm-up draws have the targets' known moments."""
import os, sys, time, json
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,"tests"))
import numpy as np
from conftest import model_spec
from parity_util import make_sampler
def run(name, C, n, check):
    stem, data = model_spec(name)
    s = make_sampler(stem, data, C, warmup=100)
    s.seed(np.arange(C, dtype=np.uint64) + 424242)
    t0 = time.time(); s.step(n); dt = time.time() - t0
    s.step(50, record_draws=True)
    d = s.draws(0, 50)
    th, st = s.get("theta"), s.get("status")
    ok = bool(np.isfinite(th).all() and not st.any())
    print(f"{name:14s} C={C:7d} n={n:5d}  {C*n/dt:.3e} it/s  finite&status-clear={ok}  acc={s.get('acc').mean():.3f}  {check(d)}", flush=True)
    return ok
ok = True
ok &= run("funnel10", 65536, 3000, lambda d: "v: mean %+.3f sd %.3f (target 0, 3)" % (d[..., 0].mean(), d[..., 0].std()))
ok &= run("illnormal100", 16384, 1000, lambda d: "sd ratio to s_i = i/10: min %.3f max %.3f" % tuple(np.quantile(d.reshape(-1, 100).std(axis=0) / (np.arange(1, 101) / 10.0), [0, 1])))
ok &= run("funnel100", 131072, 300, lambda d: "v: mean %+.3f sd %.3f" % (d[..., 0].mean(), d[..., 0].std()))
ok &= run("glmm45", 8192, 1500, lambda d: "alpha mean %.3f, log sigma mean %.3f" % (d[..., 0].mean(), d[..., 4].mean()))
print("soak", "OK" if ok else "FAILED")

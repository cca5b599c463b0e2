"""dev check of klhr_free_kernel against klhr_iter_kernel: python tools/free_dev.py [model] [C] [n]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import model_spec
from parity_util import make_sampler
name = sys.argv[1] if len(sys.argv) > 1 else "glmm45"
C = int(sys.argv[2]) if len(sys.argv) > 2 else 21
n = int(sys.argv[3]) if len(sys.argv) > 3 else 60
stem, data = model_spec(name)
seeds = np.arange(C, dtype=np.uint64) + 8800
res = {}
for mode, mh in (("octet", ""), ("free", "exact"), ("free", "")):
    os.environ["KLHR_KERNEL"] = mode
    os.environ["KLHR_FREE_MH"] = mh
    s = make_sampler(stem, data, C, warmup=100)
    s.seed(seeds)
    t0 = time.perf_counter()
    s.step(n, record_draws=True)
    dt = time.perf_counter() - t0
    res[(mode, mh)] = (s.draws(0, n), s.get("rng"), s.get("pca_v"), s.get("grad_evals"), s.get("status"))
    print(mode, mh, s.kernel_name(), f"{dt*1e3:.1f} ms", "status", np.unique(res[(mode, mh)][4]), flush=True)
a, b, c = res[("octet", "")], res[("free", "exact")], res[("free", "")]
for k, (x, y) in enumerate(zip(a, b)):
    print("exact field", k, "equal" if np.array_equal(x, y, equal_nan=True) else f"DIFF max {np.nanmax(np.abs(x.astype(float) - y.astype(float)))}")
if not np.array_equal(a[0], b[0], equal_nan=True):
    bad = np.argwhere(~np.isclose(a[0], b[0], rtol=0, atol=0, equal_nan=True))
    print("first differing (iter, chain, coord):", bad[:5])
print("prod vs octet max |diff| draws", np.nanmax(np.abs(a[0] - c[0])), "rng equal", np.array_equal(a[1], c[1]))

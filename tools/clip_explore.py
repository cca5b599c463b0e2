"""How long do free-running chains with a small clip bound (scale_clip = grad_clip = 3: the closed form's exact patch is the
common path) follow the CPU oracle?   python tools/clip_explore.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import model_spec
from parity_util import make_sampler
from oracle import oracle as O
O.build()
C, n = 24, 49
for name in ("funnel10", "funnel100", "illnormal100", "ar1x100"):
    stem, data = model_spec(name)
    seeds = np.arange(C, dtype=np.uint64) + 3100
    ref = np.array([O.Chain(stem, data, seed=int(sd), warmup=100, scale_clip=3.0, grad_clip=3.0).draw_many(n) for sd in seeds]).transpose(1, 0, 2)
    forms = [("auto", {})] if name == "funnel10" else [("fit12", {"KLHR_KERNEL": "split", "KLHR_FITN": "0", "KLHR_FUSED": "0"}), ("fitn", {"KLHR_KERNEL": "split", "KLHR_FITN": "1", "KLHR_FUSED": "0"}),
                                                      ("tile", {"KLHR_KERNEL": "split"}), ("octet", {"KLHR_KERNEL": "octet"})]
    if name == "funnel10":
        forms += [("octet", {"KLHR_KERNEL": "octet"}), ("split-tile", {"KLHR_KERNEL": "split"})]
    for form, env in forms:
        for k in ("KLHR_KERNEL", "KLHR_FITN", "KLHR_FUSED"):
            os.environ.pop(k, None)
        os.environ.update(env)
        s = make_sampler(stem, data, C, warmup=100, scale_clip=3.0, grad_clip=3.0)
        s.seed(seeds)
        s.step(n, record_draws=True)
        got = s.draws(0, n)
        err = np.abs(got - ref).max(axis=2) / 1.0          # (draw, chain)
        first = np.array([np.argmax(err[:, c] > 1e-9) if (err[:, c] > 1e-9).any() else n for c in range(C)])
        print(f"{name:13s} {form:10s} {s.kernel_name():40s} chains equal to 1e-9 for all {n} draws: {(first == n).sum()}/{C}; first divergence draws {sorted(first[first < n])[:8]}; "
              f"max err over agreeing prefix {max(err[:first[c], c].max() if first[c] else 0 for c in range(C)):.1e}", flush=True)

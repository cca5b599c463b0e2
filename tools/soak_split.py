"""Soak of the split pipeline's mappings: several seeds x configurations x flags, 400 draws with warmup = 300 (three window
closures), finite states and clear status flags expected.  python tools/soak_split.py [repeats]"""
import sys, os, time, itertools
import numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT)
from klhr_amd import _lib
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cfgs = [("funnel", 100, 4096), ("ill-normal", 100, 8192), ("funnel", 100, 16384), ("ar1", 100, 2048), ("normal", 50, 300), ("ill-normal", 30, 33),
        ("funnel", 100, 40000)]
bad = 0
t0 = time.time()
for rep in range(reps):
    for (m, D, C), (variant, over), flags in itertools.product(cfgs, ((0, False), (0, True), (1, True), (2, False)), ({}, {"eigen_method_one": True})):
        data = np.array([D-1.0]) if m == "funnel" else np.array([float(D)])
        s = _lib.Sampler(_lib.MODEL_IDS[m], data, C, warmup=300, variant=variant, overrelaxed=over, **flags)
        s.seed(np.arange(C, dtype=np.uint64) * 7 + 1000 * rep + 3)
        s.step(400); s.sync()
        th = s.get("theta"); st = s.get("status")
        ok = np.isfinite(th).all() and not (st & ~2).any()          # (bit 1: a non-finite state the reference's blend can produce)
        if not ok:
            bad += 1
            print("BAD", m, D, C, variant, over, flags, s.kernel_name(), "non-finite rows", int((~np.isfinite(th)).any(axis=1).sum()), "status", np.unique(st))
    print(f"rep {rep}: {time.time() - t0:.0f} s", flush=True)
print("soak done, bad =", bad)

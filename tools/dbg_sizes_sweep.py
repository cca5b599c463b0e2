"""Odd chain counts around the mapping thresholds of the split pipeline: finite states, clear status flags, and the first chain's
state independent of the launch size.  python tools/dbg_sizes_sweep.py [model D]"""
import sys, os
import numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT)
from klhr_amd import _lib
m = sys.argv[1] if len(sys.argv) > 1 else "funnel"; D = int(sys.argv[2]) if len(sys.argv) > 2 else 100
data = np.array([D-1.0]) if m == "funnel" else np.array([float(D)])
ref = None
for variant, over in ((0, False), (0, True), (1, False), (2, False)):
    ref = None
    for C in (1, 15, 16, 17, 31, 33, 127, 8191, 8192, 8193, 16383, 16385, 32767, 32769):
        s = _lib.Sampler(_lib.MODEL_IDS[m], data, C, warmup=100, variant=variant, overrelaxed=over)
        s.seed(np.arange(C, dtype=np.uint64) + 77); s.step(56); s.sync()
        th = s.get("theta"); ok = np.isfinite(th).all() and not s.get("status").any()
        first = (th[0].copy(), s.get("rng")[0].copy())
        same = ref is None or (np.array_equal(first[0], ref[0]) and np.array_equal(first[1], ref[1]))
        if ref is None: ref = first
        print(f"variant {variant} over {int(over)} C {C:6d} {s.kernel_name():60s} finite/status {'ok' if ok else 'BAD'} first chain {'same' if same else 'DIFFERENT'}", flush=True)

"""Time klhr_step for a model bench.py's --model does not take: python tools/time_steps.py <stem> <D> <chains> [draws]  -> us per iteration, kernels
(environment knobs such as KLHR_TILE / KLHR_FUSED apply)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from klhr_amd import _lib
stem, D, C = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]); n = int(sys.argv[4]) if len(sys.argv) > 4 else 50
s = _lib.Sampler(_lib.MODEL_IDS[stem], np.array([float(D)]), C, warmup=100)
s.seed(np.arange(C, dtype=np.uint64)); s.step(150); s.sync()
best = 1e9
for r in range(3):
    t0 = time.perf_counter(); s.step(n); s.sync(); best = min(best, time.perf_counter() - t0)
print(f"{stem}-{D} x {C}: {1e6 * best / n:.1f} us per iteration, {C * n / best:.4g} it/s, {s.kernel_name()}, tile width {os.environ.get('KLHR_TILE', 'default')}")

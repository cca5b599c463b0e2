"""BASELINE.json's configurations at FULL size, checked through properties that do not need an
oracle run of that size:
  * size independence — a chain is a pure function of (seed, configuration): chains picked out of the full-size run are
    bit-identical to the same seeds run in a small launch (whose parity with the oracle / reference the other tests pin);
  * sharding — two half-size samplers seeded with `base + global index` reproduce the full run's halves bit for bit
    (the multi-GPU layout of klhr_amd/parallel.py, here on one device);
  * sanity of the whole population — finite states, status flags clear, acceptance and work per iteration in range,
    window-closure bookkeeping identical in every chain."""
import numpy as np
import pytest

from conftest import model_spec
from parity_util import make_sampler

pytestmark = pytest.mark.gpu

# every configuration crosses the first window closure (draw 50: close_window at full occupancy), funnel-10 the second too
CONFIGS = [("funnel10", 65536, 130), ("illnormal100", 16384, 60), ("glmm45", 8192, 54), ("funnel100", 131072, 54)]
FIELDS = ["theta", "rng", "mean", "cov", "eigvecs", "eigvals", "om_m", "om_v", "pca_v", "counters", "acc", "grad_evals"]


@pytest.mark.parametrize("name,C,n", CONFIGS)
def test_full_size_population(oracle, name, C, n):
    stem, data = model_spec(name)
    base = 1000
    full = make_sampler(stem, data, C, warmup=100)
    full.seed(np.arange(C, dtype=np.uint64) + base)
    full.step(n)
    st = {f: full.get(f) for f in FIELDS}
    # ---- the population
    assert np.isfinite(st["theta"]).all()
    assert not full.get("status").any()
    cnt = st["counters"]
    assert np.all(cnt[:, 0] == n)                                   # _draw
    assert np.all(cnt[:, 1] == cnt[0, 1]) and np.all(cnt[:, 2] == cnt[0, 2]) and np.all(cnt[:, 3] == cnt[0, 3])
    acc = st["acc"]
    assert 0.2 < acc.mean() <= 1.0 and acc.min() >= 0.0          # (glmm-45 is still in its transient after ~50 draws: ~0.5)
    ge = st["grad_evals"] / n
    assert 40 < ge.mean() < 1200 and ge.min() >= 9                  # >= 1 line + 1 KL evaluation per draw
    # ---- size independence: the same seeds in small launches
    picks = [np.arange(0, 64), np.arange(C // 2 - 3, C // 2 + 5), np.array([C - 1])]
    for idx in picks:
        small = make_sampler(stem, data, len(idx), warmup=100)
        small.seed(idx.astype(np.uint64) + base)
        small.step(n)
        for f in FIELDS:
            assert np.array_equal(small.get(f), st[f][idx]), (f, idx[:3])
    # ---- sharding: two halves seeded base + global index
    if C <= 65536:
        h = C // 2
        for r in range(2):
            sh = make_sampler(stem, data, h, warmup=100)
            sh.seed(np.arange(r * h, (r + 1) * h, dtype=np.uint64) + base)
            sh.step(n)
            for f in ("theta", "rng", "acc", "grad_evals"):
                assert np.array_equal(sh.get(f), st[f][r * h:(r + 1) * h]), (f, r)


@pytest.mark.parametrize("name", ["funnel10", "illnormal100", "glmm45", "earnings"])
def test_odd_chain_counts_every_variant(name):
    """Partially filled wavefronts (1, 9, 65 chains) for every sampler variant with and without over-relaxation: states stay
    finite, status flags clear, and the first chain does not depend on how many others ran beside it.  (tools/stress.py
    is the larger, process-isolated version of this net.)"""
    stem, data = model_spec(name)
    n = 40 if name == "glmm45" else 110
    for variant in (0, 1, 2, 3):
        for over in ((False,) if variant == 3 else (False, True)):
            first = None
            for C in (1, 9, 65):
                kw = dict(warmup=100, variant=variant, overrelaxed=over)
                if variant == 3:
                    kw["tol"] = 1e-12
                s = make_sampler(stem, data, C, **kw)
                s.seed(np.arange(C, dtype=np.uint64) + 17)
                s.step(n)
                th = s.get("theta")
                assert np.isfinite(th).all() and not s.get("status").any(), (name, variant, over, C)
                if first is None:
                    first = th[0].copy()
                else:
                    assert np.array_equal(th[0], first), (name, variant, over, C)


@pytest.mark.parametrize("name", ["funnel100", "ar1x100", "funnel10"])
def test_mapping_thresholds_do_not_show_in_a_chain(name):
    """The split pipeline switches mappings with the chain count: whole iterations per 16-chain tile (quad mode: four lanes per
    chain) in one launch up to 16 384 chains — the one-wavefront-per-SIMD build up to 8 192, the two-per-SIMD build above —,
    64-chain tiles (lane = chain) above; the node-parallel fit kernel up to 32 768 chains, the lane = chain fit kernels above.
    Dimension <= 26 (funnel10): the tile kernel up to 16 384 chains, the lane-per-chain kernel from 16 385 on.
    The first chains' states after 60 draws (first window closure included) are the same bits on both sides of each threshold
    and in a small launch."""
    stem, data = model_spec(name)
    n, base = 60, 31000
    ref = None
    for C in (40, 8192, 8193, 16384, 16385, 32768, 32769):
        s = make_sampler(stem, data, C, warmup=100)
        s.seed(np.arange(C, dtype=np.uint64) + base)
        s.step(n)
        assert not s.get("status").any(), C
        st = [s.get(f)[:40] for f in ("theta", "rng", "cov", "eigvecs", "pca_v", "acc", "grad_evals", "counters")]
        if ref is None:
            ref = st
        else:
            for a, b in zip(ref, st):
                assert np.array_equal(a, b), (C, s.kernel_name())


def test_cfg5_pooled_shard_full_size_through_rccl(kernel_mode):
    """BASELINE config 5's per-GPU shard at full size in POOLED mode: funnel-100D, 131 072 chains, pooled="full" with the collective
    inside the library (klhr_comm_init with a one-rank communicator: ncclAllReduce of 1 + D + D^2 doubles on the handle's
    stream) through BOTH window closures of warmup = 100 — the LDS-tiled MFMA Gram kernel over 49 x 131 072 and 99 x 131 072
    draws.  Every chain carries the same pooled mean / diagonal / eigenvectors afterwards, and the pooled mean is the window's
    column sum over its row count as the all-reduced statistics hold them."""
    if kernel_mode != "auto":
        pytest.skip("the production mapping of this configuration (split pipeline, 64-chain tiles)")
    from klhr_amd import BSModel
    from klhr_amd.parallel import ShardedKLHRSINH
    import os
    from conftest import STAN
    m = BSModel(os.path.join(STAN, "funnel.stan"), data={"D": 99})
    assert m.dim() == 100
    C = 131072
    sh = ShardedKLHRSINH(m, C, seed=77, pooled="full", rccl=True, warmup=100)
    s = sh.local._sampler
    assert s.comm_world() == 1 and "stream" in s.kernel_name()
    sh.advance(49)
    s.pool_allreduce()
    tot = s.pool_fetch()
    assert tot[0] == 49 * C
    sh.advance(1)                      # first closure: pooled values installed
    mean, cov, evec = s.get("mean"), s.get("cov"), s.get("eigvecs")
    assert np.all(mean == mean[0]) and np.all(cov == cov[0]) and np.all(evec == evec[0])
    assert np.allclose(mean[0], tot[1:101] / tot[0], rtol=1e-12, atol=1e-14)
    sh.advance(60)                     # second closure (draw 100) and ten draws beyond it
    mean2, cov2 = s.get("mean"), s.get("cov")
    assert np.all(mean2 == mean2[0]) and np.all(cov2 == cov2[0]) and not np.array_equal(mean2[0], mean[0])
    assert np.isfinite(s.get("theta")).all() and not s.get("status").any()
    assert np.all(s.get("counters")[:, 0] == 110)

"""Oracle restatement of the reference's Slice sampler (slice.py: Neal's univariate slice sampler — stepping out with
m = inf, shrinkage — along the KLHR direction draw) vs golden G13, which traces the reference's own
draw().  Also pins Generator.exponential (ziggurat) and Generator.uniform(low, high) against NumPy itself."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden, model_spec

FILES = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "g13_slice_*_w*.npz")))


def parse(fn):
    parts = fn[len("g13_slice_"):-4].split("_")
    flags = {}
    if "m2" in parts[3:]:
        flags["eigen_method_one"] = False
    if "s" in parts[3:]:
        flags["scale_dir_cov"] = True
    return parts[0], flags


def test_exponential_and_ranged_uniform_streams_equal_numpy(oracle):
    for seed in (0, 1, 987654321):
        a, st = oracle.rng_exponential(oracle.seed_state(seed), 300000)
        g = np.random.default_rng(seed)
        assert np.array_equal(a, g.exponential(size=300000))          # includes tail (idx 0) and wedge rejections
        v, st = oracle.rng_uniform_range(st, -3.25, 11.5)
        assert v == g.uniform(low=-3.25, high=11.5)
        assert np.array_equal(oracle.rng_normal(st, 5)[0], g.normal(size=5))


@pytest.mark.parametrize("fn", FILES)
def test_slice_teacher_forced(oracle, fn):
    name, flags = parse(fn)
    g = golden(fn)
    stem, data = model_spec(name)
    M, warmup, seed = int(g["M"][0]), int(g["warmup"][0]), int(g["seed"][0])
    c = oracle.SliceChain(stem, data, seed=seed, warmup=warmup, **flags)
    assert np.array_equal(c.get("rng"), g["state_before"][0]) and np.array_equal(c.get("theta"), g["theta0"])
    flips = 0
    for m in range(M):
        c.set("theta", g["theta_before"][m]); c.set("rng", g["state_before"][m])
        rho = c.random_direction()
        # the direction is exact until the first window closure; afterwards it inherits CCIPCA's irreproducible second
        # component, so it is compared loosely and the reference's rho drives the slice step
        tol = 1e-13 if m < 50 else 2e-3
        assert np.allclose(rho, g["rho"][m], rtol=tol, atol=tol * 1e-1), m
        e0 = c.slice_evals()
        c.set("draw", m)
        th = c.uni_slice(g["rho"][m])
        same = np.array_equal(c.get("rng"), g["state_after"][m]) and c.slice_evals() - e0 == g["evals"][m]
        flips += not same
        if same:
            assert np.allclose(th, g["theta_after"][m], rtol=1e-13, atol=1e-14), m
            assert abs(c.get("acc") - g["accprob"][m]) < 1e-15
        c.set("theta", g["theta_after"][m]); c.set("rng", g["state_after"][m])
        c.adapt()
        if m + 1 in g["snap_draws"]:
            k = m + 1
            for f in ("mean", "cov", "om_m", "om_v", "omd_m", "omd_v"):
                assert np.allclose(c.get(f), g[f"snap{k}_{f}"], rtol=1e-12, atol=1e-13), (f, k)
            ev, rv = c.get("eigvecs"), g[f"snap{k}_eigvecs"]
            assert np.allclose(ev[:, 0], rv[:, 0], rtol=1e-9, atol=1e-12), k
            assert np.allclose(ev[:, 1:], rv[:, 1:], rtol=0, atol=2e-2), k
            assert np.allclose(c.get("eigvals")[0], g[f"snap{k}_eigvals"][0], rtol=1e-9)
    # a comparison g(x) <= log y decided by the last bits of the log density (NumPy pairwise sums vs the oracle's
    # sequential ones) may go the other way: allowed in < 0.5 % of the draws
    assert flips <= 0.005 * M + 1, (fn, flips)


def test_slice_free_running_before_first_closure(oracle):
    """Until draw 50 nothing depends on CCIPCA: the oracle follows the reference's own draw() loop step for step."""
    for fn in ("g13_slice_funnel10_s0_w100.npz", "g13_slice_normal2_s0_w100.npz", "g13_slice_illnormal100_s0_w100.npz"):
        name, flags = parse(fn)
        g = golden(fn)
        stem, data = model_spec(name)
        c = oracle.SliceChain(stem, data, seed=int(g["seed"][0]), warmup=100, **flags)
        th = np.array([c.draw() for _ in range(50)])
        assert np.allclose(th, g["theta_after"][:50], rtol=1e-12, atol=1e-13), fn
        assert np.array_equal(c.get("rng"), g["state_after"][49]), fn
        assert c.slice_evals() == int(g["evals"][:50].sum())

"""Over-relaxed proposal (klhr_sinh.py:131-137,215-228).  The reference draws its binomial/beta
variates from SciPy's global unseeded generator, so only the distribution can be pinned: the oracle (which shares this
build's RNG convention with the HIP kernels) is compared with samples of the reference's own `_overrelaxed_proposal`
(fixture G10) by two-sample Kolmogorov-Smirnov tests, and its Phi / Phi^-1 with scipy.special."""
import numpy as np
import scipy.special as sp
import scipy.stats as st

from conftest import golden


def test_ndtr_ndtri_vs_scipy(oracle):
    L = oracle.lib()
    ps = np.concatenate([np.logspace(-300, -1, 200), np.linspace(0.01, 0.99, 199), 1 - np.logspace(-16, -2, 50)])
    mine = np.array([L.klhr_oracle_ndtri(float(p)) for p in ps])
    ref = sp.ndtri(ps)
    assert np.all(np.abs(mine - ref) <= 1e-14 * np.maximum(1e-3, np.abs(ref)))
    xs = np.linspace(-37, 8, 400)
    a = np.array([L.klhr_oracle_ndtr(float(x)) for x in xs])
    assert np.all(np.abs(a - sp.ndtr(xs)) <= 1e-13 * sp.ndtr(xs))
    assert L.klhr_oracle_ndtri(0.0) == -np.inf and L.klhr_oracle_ndtri(1.0) == np.inf


def oracle_zp_samples(oracle, eta, n, seed):
    c = oracle.Chain("normal", {"D": 2}, seed=seed, overrelaxed=True)
    rho = np.array([1.0, 0.0])
    out = np.empty(n)
    th0 = c.get("theta").copy()
    for k in range(n):
        c.set("theta", th0)
        c.metropolis(eta, rho)
        out[k] = c.get("diag")["zp"]
    return out


def test_overrelaxed_distribution_matches_reference(oracle):
    g = golden("g10_overrelaxed.npz")
    assert int(g["K"][0]) == 10
    for i, eta in enumerate(g["etas"]):
        mine = oracle_zp_samples(oracle, eta, 20000, seed=100 + i)
        ref = g[f"zp{i}"]
        # r == K - r leaves u' = u, i.e. an atom at z' = T(T_inv(0)) = 0 up to rounding: compare its mass, then the rest
        am, ar = np.abs(mine) < 1e-9, np.abs(ref) < 1e-9
        pm, pr = am.mean(), ar.mean()
        assert abs(pm - pr) < 4 * np.sqrt(pr * (1 - pr) / ref.size + 1e-6), (i, pm, pr)
        ks = st.ks_2samp(mine[~am], ref[~ar])
        assert ks.pvalue > 1e-3, (i, ks)
        # over-relaxation pushes z' to the other side of the proposal's median from 0
        u = float(g[f"u{i}"][0])
        z0 = 0.0
        med = np.median(mine)
        assert (np.mean(mine > z0) > 0.5) == (u < 0.5) or abs(u - 0.5) < 0.05, (u, med)


def test_overrelaxed_chain_runs_and_targets_normal(oracle):
    c = oracle.Chain("normal", {"D": 2}, seed=3, warmup=100, overrelaxed=True)
    th = c.draw_many(3000)[500:]
    assert np.abs(th.mean(axis=0)).max() < 0.15 and np.abs(th.var(axis=0) - 1).max() < 0.2

"""Oracle RNG vs golden G2 (NumPy PCG64 + ziggurat stream the reference consumes: mcmc.py:9-15)."""
import numpy as np

from conftest import golden


def test_seedsequence_and_streams_bit_exact(oracle):
    g = golden("g2_rng.npz")
    for s in g["seeds"]:
        s = int(s)
        st = oracle.seed_state(s)
        assert np.array_equal(st, g[f"s{s}_state0"])
        raw, _ = oracle.rng_raw(st, 4096)
        assert np.array_equal(raw, g[f"s{s}_raw"])
        d, _ = oracle.rng_double(st, 4096)
        assert np.array_equal(d, g[f"s{s}_double"])
        z, st1 = oracle.rng_normal(st, 4096)
        assert np.array_equal(z, g[f"s{s}_normal"])
        u, st2 = oracle.rng_uniform(st1)
        assert u == g[f"s{s}_uniform_after"][0]
        assert np.array_equal(st2, g[f"s{s}_state_after"])


def test_long_normal_stream_with_tail_and_wedge(oracle):
    g = golden("g2_rng.npz")
    z, st = oracle.rng_normal(g["long_state0"], 200_000)
    assert np.array_equal(st, g["long_state_after"])
    assert np.array_equal(z[::997], g["long_every"])
    assert len(g["long_tail_idx"]) > 10                     # the idx==0 tail path was exercised
    assert np.array_equal(z[g["long_tail_idx"]], g["long_tail_val"])
    assert np.allclose([z.sum(), (z * z).sum(), np.abs(z).max()], g["long_sum"], rtol=1e-13)


def test_seedsequence_matches_numpy_for_wide_seeds(oracle):
    M = (1 << 64) - 1
    for s in [7, 2**32 - 1, 2**32, 2**32 + 5, 2**63 + 11, 987654321987, M]:
        r = np.random.default_rng(s).bit_generator.state["state"]
        assert list(oracle.seed_state(s)) == [r["state"] >> 64, r["state"] & M, r["inc"] >> 64, r["inc"] & M]


def test_scaled_normals(oracle):
    g = golden("g2_rng.npz")
    z, st = oracle.rng_normal(g["scale_state0"], 21)
    assert np.array_equal(0.0 + 0.1 * z[:10], g["scale_a"])        # rng.normal(scale=0.1,size=D)  mcmc.py:15
    assert np.array_equal((0.0 + 1.0 * z[10:20]) * 0.1, g["scale_b"])   # rng.normal(size=D)*0.1   klhr_sinh.py:90
    assert (0.0 + 1.0 * z[20]) * 0.1 == g["scale_c"][0]


def test_multivariate_normal_rank_rule(oracle):
    """Generator.multivariate_normal(m, diag(c)) == m + sqrt(c) * z[rank]  (klhr_sinh.py:211-212)."""
    g = golden("g2_rng.npz")
    for i in range(int(g["mvn_n"][0])):
        m, c, x = g[f"mvn{i}_m"], g[f"mvn{i}_c"], g[f"mvn{i}_x"]
        D = m.size
        z, st = oracle.rng_normal(g[f"mvn{i}_state0"], D)
        assert np.array_equal(st, g[f"mvn{i}_state1"])
        order = np.argsort(-c, kind="stable")
        rank = np.empty(D, dtype=int)
        rank[order] = np.arange(D)
        mine = m + np.sqrt(c) * z[rank]
        if D <= 25 or np.all(c == c[0]):
            assert np.array_equal(mine, x), i
        else:       # LAPACK gesdd path for D >= 26 perturbs singular values by <= 1 ulp
            scale = np.abs(m) + np.abs(np.sqrt(c) * z[rank])
            assert np.all(np.abs(mine - x) <= 4.5e-16 * scale), i


def test_choice_with_p(oracle):
    g = golden("g2_rng.npz")
    p = g["choice_p"]
    cdf = np.cumsum(p)
    cdf /= cdf[-1]
    u, st = oracle.rng_double(g["choice_state0"], 256)
    assert np.array_equal(np.searchsorted(cdf, u, side="right"), g["choice_j"])
    assert np.array_equal(st, g["choice_state1"])

"""Oracle KLHRSINH chain vs golden G5/G6/G8 (teacher-forced per iteration + free-running windows) and
MH vs G9.  Tolerances: SURVEY.md §8c (BFGS trajectories cannot be bit-matched: OpenBLAS FMA dot, libm)."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden, model_spec

FILES = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "g8_*_w*.npz")))


def parse(fn):
    parts = fn[3:-4].split("_")
    flags = {}
    if parts[-1] == "e":
        flags["eigen_method_one"] = True
    if parts[-1] == "s":
        flags["scale_dir_cov"] = True
    return parts[0], flags


@pytest.mark.parametrize("fn", FILES)
def test_teacher_forced_pieces(oracle, fn):
    name, flags = parse(fn)
    g = golden(fn)
    stem, data = model_spec(name)
    M, warmup, seed = int(g["M"][0]), int(g["warmup"][0]), int(g["seed"][0])
    c = oracle.Chain(stem, data, seed=seed, warmup=warmup, **flags)
    assert np.array_equal(c.get("rng"), g["state_before"][0])      # constructor + _initialize consumed the same stream
    assert np.array_equal(c.get("theta"), g["theta0"])
    glmm = name.startswith("glmm")
    n_count_mismatch = 0
    n_eta_far = 0
    snaps = set(int(m) for m in g["snap_draws"])
    for m in range(M):
        c.set("theta", g["theta_before"][m])
        c.set("rng", g["state_before"][m])
        # --- a5 direction (bit-exact while D <= 25; ulp-level otherwise)
        rho = c.random_direction()
        # before the first window closes m=0, cov=1: pure RNG + normalisation.  Afterwards rho inherits the
        # reference's own CCIPCA irreproducibility (see snapshot tolerance below).
        rtol = 1e-13 if m < 50 else 2e-3
        assert np.allclose(rho, g["rho"][m], rtol=rtol, atol=rtol * 1e-1), m
        # --- a10 fit, from the reference's rho
        eta = c.fit(g["rho"][m])
        d = c.get("diag")
        assert d["x0"] == g["x0"][m]
        assert np.array_equal([d["init2"], d["init3"]], g["init"][m][2:]), m
        assert np.array_equal(c.get("rng"), g["state_mh"][m]), m
        same_counts = d["nfev1"] == g["nfev1"][m] and d["nfev2"] == g["nfev2"][m]
        failed = g["status1"][m] != 0 or g["status2"][m] != 0
        if not failed:
            # converged searches: identical evaluation counts (and status) in >= 99 % of records
            agree = same_counts and d["status1"] == 0 and d["status2"] == 0
            n_count_mismatch += (not agree)
            same_counts = agree
        if same_counts or failed:
            # (a failed final line search ends at rounding-noise level: its evaluation COUNT is noise, its x is not;
            #  on targets whose KL is nearly flat in (d, e) — Gaussian ar1 — a converged fit can also stop a step apart)
            tol = 1e-5 if failed else 2e-6
            close = np.allclose(eta, g["eta"][m], rtol=tol, atol=tol)
            n_eta_far += (not close)
            assert np.allclose(eta, g["eta"][m], rtol=5e-2, atol=5e-2) or failed, (m, eta, g["eta"][m])
            assert abs(d["mode"] - g["mode"][m]) <= 1e-6 * (1 + abs(g["mode"][m])), m
        # --- a12 MH step, from the reference's eta and rho
        c.set("draw", m)
        c.metropolis(g["eta"][m], g["rho"][m])
        d = c.get("diag")
        assert np.isclose(d["zp"], g["zp"][m], rtol=1e-11, atol=1e-12), m
        assert np.isclose(d["r"], g["r"][m], rtol=1e-9, atol=1e-9), m
        assert d["logu"] == pytest.approx(g["logu"][m], rel=1e-15)
        if abs(g["logu"][m] - min(0.0, g["r"][m])) > 1e-8:
            assert bool(d["accept"]) == bool(g["accept"][m]), m
            assert np.allclose(c.get("theta"), g["theta_after"][m], rtol=1e-11, atol=1e-12), m
        assert np.array_equal(c.get("rng"), g["state_after"][m]), m
        # --- a14-a16 closure-or-update, from the reference's theta
        c.set("theta", g["theta_after"][m])
        c.adapt()
        if (m + 1) in snaps:
            k = m + 1
            for f in ["mean", "cov", "eigvecs", "eigvals", "om_m", "om_v", "pca_v"] + (["omd_m", "omd_v"] if flags.get("scale_dir_cov") else []):
                ref = g[f"snap{k}_{f}"]
                # CCIPCA's 2nd vector collapses to ~1e-25 at n = l+1 (weight 0, onlinepca.py:20) and regrows from
                # rounding noise: the reference itself is only reproducible to ~1e-7 there (measured vs longdouble)
                mine = c.get(f)
                if f in ("pca_v", "eigvecs", "eigvals"):
                    lead = (Ellipsis, 0)
                    assert np.allclose(mine[lead], ref[lead], rtol=1e-9, atol=1e-12), (k, f, "leading component")
                    assert np.allclose(mine, ref, rtol=2e-3, atol=2e-3 * np.abs(ref).max()), (k, f)
                else:
                    assert np.allclose(mine, ref, rtol=1e-9, atol=1e-12), (k, f)
            assert c.get("om_N") == g[f"snap{k}_om_N"] and c.get("pca_n") == g[f"snap{k}_pca_n"]
    ok_frac = 1.0 - n_count_mismatch / M
    # (the AR(1) target is Gaussian: the KL objective is nearly flat in the tail parameters and BFGS' stopping step is
    #  decided by rounding more often — 3 % of records here, against 0 % on funnel / ill-normal / normal)
    assert ok_frac >= (0.95 if name == "ar1x100" else 0.985), (fn, n_count_mismatch, M)
    assert n_eta_far <= 0.02 * M + (0.15 * M if name in ("glmm45", "earnings") else 0), (fn, n_eta_far, M)


@pytest.mark.parametrize("fn", [f for f in FILES if "funnel10_s" in f or "normal2" in f])
def test_free_running_window(oracle, fn):
    name, flags = parse(fn)
    g = golden(fn)
    stem, data = model_spec(name)
    M, warmup, seed = int(g["M"][0]), int(g["warmup"][0]), int(g["seed"][0])
    c = oracle.Chain(stem, data, seed=seed, warmup=warmup, **flags)
    th = c.draw_many(M)
    err = np.abs(th - g["theta_after"]).max(axis=1)
    assert err[:50].max() < 1e-9, err[:50].max()
    # afterwards ulp-level noise is amplified chaotically: compare until first divergence
    bad = np.flatnonzero(err > 1e-6)
    first = int(bad[0]) if bad.size else M
    assert first >= 50
    acc_o = np.any(th[1:first] != th[:first - 1], axis=1)
    acc_r = g["accept"][1:first]
    assert np.array_equal(acc_o, acc_r)


def test_public_api_fixture_and_counters(oracle):
    g = golden("g8_public_api.npz")
    for seed in (0, 1):
        c = oracle.Chain("funnel", {"D": 9}, seed=seed, warmup=100)
        th = c.draw_many(60)
        assert np.allclose(th, g[f"s{seed}_thetas"][:60], atol=1e-9)
    g0 = golden("g8_funnel10_s0_w100.npz")
    c = oracle.Chain("funnel", {"D": 9}, seed=0, warmup=100)
    c.draw_many(50)
    assert c.get("grad_evals") == g0["grad_evals"][49]
    assert c.get("acc") == pytest.approx(g0["accprob"][49], abs=1e-15)


@pytest.mark.parametrize("name", ["normal2", "normal100"])
def test_mh_sampler(oracle, name):
    g = golden("g9_mh.npz")
    stem, data = model_spec(name)
    for seed in (0, 1):
        c = oracle.Chain(stem, data)
        c.mh_seed(seed)
        assert np.array_equal(c.get("theta"), g[f"{name}_s{seed}_theta0"])
        th = c.mh_draw(0.09, 2000)
        ref = g[f"{name}_s{seed}_thetas"]
        mine = th[::10] if name == "normal100" else th
        # accept decisions compare log(u) with a difference of two log densities: identical unless |margin| ~ 1e-14
        assert np.allclose(mine, ref, rtol=0, atol=1e-12)
        assert np.array_equal(c.get("rng"), g[f"{name}_s{seed}_state1"])
        assert c.get("acc") == pytest.approx(g[f"{name}_s{seed}_acc"][0], abs=1e-12)

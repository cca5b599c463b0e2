"""Oracle sinh-arcsinh primitives, KL objective, line objective and models vs golden G1/G3/G4
(klhr_sinh.py:78-84,100-180,230-240)."""
import numpy as np
import pytest

from conftest import golden, model_spec

RTOL = 1e-12


def close(a, b, rtol=RTOL, atol=0.0):
    a, b = np.asarray(a), np.asarray(b)
    both_nan = np.isnan(a) & np.isnan(b)
    same_inf = np.isinf(a) & np.isinf(b) & (np.sign(a) == np.sign(b))
    fin = np.isfinite(a) & np.isfinite(b)
    ok = both_nan | same_inf | (fin & (np.abs(a - b) <= atol + rtol * np.abs(b)))
    return bool(np.all(ok))


def test_quadrature_matches_reference(oracle):
    g = golden("g1_quadrature.npz")
    x, w = oracle.quadrature(8)
    assert np.array_equal(x, g["x"]) and np.array_equal(w, g["w"])
    assert abs(w.sum() - 1.0) < 1e-15


def test_primitives_grid(oracle):
    g = golden("g3_primitives.npz")
    c = oracle.Chain("normal", {"D": 2}, seed=0)
    for i, eta in enumerate(g["etas"]):
        for j, x in enumerate(g["xs"]):
            p = c.primitives(x, eta)
            # atol terms: the value is a difference of O(1..300) quantities in the clip cases
            assert close(p["T"], g["T"][i, j], atol=1e-13), (i, j)
            assert close(p["log_abs_jac"], g["log_abs_jac"][i, j], atol=1e-12), (i, j)
            assert close(p["grad_T"], g["grad_T"][i, j], atol=1e-13), (i, j)
            assert close(p["grad_log_abs_jac"], g["grad_log_abs_jac"][i, j], atol=1e-13), (i, j)
        for j, y in enumerate(g["ys"]):
            p = c.primitives_inv(y, eta)
            assert close(p["T_inv"], g["T_inv"][i, j], rtol=1e-11, atol=1e-13), (i, j)
            assert close(p["log_q"], g["log_q"][i, j], rtol=1e-11, atol=1e-11), (i, j)


@pytest.mark.parametrize("name", ["funnel10", "illnormal100", "glmm45", "normal2", "funnel100", "ar1x100", "earnings"])
def test_kl_and_models(oracle, name):
    g = golden("g4_kl.npz")
    stem, data = model_spec(name)
    c = oracle.Chain(stem, data, seed=11)
    th, rho, eta = g[f"{name}_theta"], g[f"{name}_rho"], g[f"{name}_eta"]
    for i in range(th.shape[0]):
        lp, grad = c.log_density_gradient(th[i])
        assert close(lp, g[f"{name}_lp"][i]), i
        assert close(grad, g[f"{name}_lpgrad"][i], atol=1e-13), i
        c.set("theta", th[i])
        v, gr = c.KL(eta[i], rho[i])
        scale = max(1.0, abs(g[f"{name}_kl"][i])) if np.isfinite(g[f"{name}_kl"][i]) else 1.0
        assert close(v, g[f"{name}_kl"][i], atol=1e-12 * scale), i
        gs = np.abs(g[f"{name}_klgrad"][i]).max()
        assert close(gr, g[f"{name}_klgrad"][i], atol=1e-12 * (gs if np.isfinite(gs) else 1.0)), i
        f, df = c.logp_grad_rho(eta[i, 0], rho[i])
        assert close([f, df], g[f"{name}_line"][i], atol=1e-12 * max(1.0, np.abs(g[f"{name}_line"][i]).max())), i


@pytest.mark.parametrize("name", ["funnel10", "illnormal100", "glmm45", "normal2", "funnel100", "ar1x100", "earnings"])
def test_kl_with_the_gradient_clip_active(oracle, name):
    """G4c: the reference's KL(eta, rho) with scale_clip = 3 — the element-wise clip of _logp_grad (klhr_sinh.py:158-161) bites at
    nearly every quadrature node, on every target (ar1 included: G4 has none there)."""
    g = golden("g4c_kl_clip.npz")
    stem, data = model_spec(name)
    c = oracle.Chain(stem, data, seed=11, scale_clip=float(g["scale_clip"][0]))
    th, rho, eta = g[f"{name}_theta"], g[f"{name}_rho"], g[f"{name}_eta"]
    assert (g[f"{name}_nclip"] > 0).sum() >= 59
    for i in range(th.shape[0]):
        c.set("theta", th[i])
        v, gr = c.KL(eta[i], rho[i])
        scale = max(1.0, abs(g[f"{name}_kl"][i])) if np.isfinite(g[f"{name}_kl"][i]) else 1.0
        assert close(v, g[f"{name}_kl"][i], atol=1e-12 * scale), i
        gs = np.abs(g[f"{name}_klgrad"][i]).max()
        assert close(gr, g[f"{name}_klgrad"][i], atol=1e-12 * (gs if np.isfinite(gs) else 1.0)), i


@pytest.mark.parametrize("name", ["funnel10", "illnormal100", "glmm45", "normal2", "ar1x100"])
def test_model_gradients_finite_difference(oracle, name):
    """The closed forms stand in for BridgeStan (parity unpinned there): at least d/dtheta is self-consistent."""
    stem, data = model_spec(name)
    c = oracle.Chain(stem, data, seed=1)
    rng = np.random.default_rng(5)
    th = rng.normal(size=c.D) * 0.3
    lp, g = c.log_density_gradient(th)
    for i in range(c.D):
        e = np.zeros(c.D)
        h = 1e-6
        e[i] = h
        fd = (c.log_density_gradient(th + e)[0] - c.log_density_gradient(th - e)[0]) / (2 * h)
        assert abs(fd - g[i]) <= 1e-6 * max(1.0, abs(g[i])), (i, fd, g[i])


def test_reference_fd_identities(oracle):
    """klhr_sinh.py:315-349 self-checks restated: grad_T, grad_log_abs_jac and the KL gradient vs central differences."""
    stem, data = model_spec("funnel10")
    c = oracle.Chain(stem, data, seed=3)
    rng = np.random.default_rng(0)
    rho = rng.normal(size=c.D)
    rho /= np.linalg.norm(rho)
    eta = rng.normal(size=4) * 0.1
    h = 1e-6
    p0 = c.primitives(0.3, eta)
    gT, gJ, gK = np.zeros(4), np.zeros(4), np.zeros(4)
    for k in range(4):
        e = np.zeros(4)
        e[k] = h
        gT[k] = (c.primitives(0.3, eta + e)["T"] - c.primitives(0.3, eta - e)["T"]) / (2 * h)
        gJ[k] = (c.primitives(0.3, eta + e)["log_abs_jac"] - c.primitives(0.3, eta - e)["log_abs_jac"]) / (2 * h)
        gK[k] = (c.KL(eta + e, rho)[0] - c.KL(eta - e, rho)[0]) / (2 * h)
    assert np.allclose(gT, p0["grad_T"], rtol=1e-6, atol=1e-8)
    assert np.allclose(gJ, p0["grad_log_abs_jac"], rtol=1e-6, atol=1e-8)
    assert np.allclose(gK, c.KL(eta, rho)[1], rtol=1e-5, atol=1e-7)

"""Oracle restatement of SUBKLHRSINH (sub_klhr_sinh.py: 3-parameter sinh-arcsinh proposal, d == 1, gradient clip at
grad_clip) vs golden G12: teacher-forced pieces and free-running windows."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden, model_spec

FILES = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "g12_sub_*_w*.npz")))


@pytest.mark.parametrize("fn", FILES)
def test_sub_teacher_forced(oracle, fn):
    name = fn[len("g12_sub_"):-4].split("_")[0]
    g = golden(fn)
    stem, data = model_spec(name)
    M, warmup, seed = int(g["M"][0]), int(g["warmup"][0]), int(g["seed"][0])
    c = oracle.Chain(stem, data, seed=seed, warmup=warmup, variant="sub")
    assert np.array_equal(c.get("rng"), g["state_before"][0]) and np.array_equal(c.get("theta"), g["theta0"])
    mism = 0
    for m in range(M):
        c.set("theta", g["theta_before"][m]); c.set("rng", g["state_before"][m])
        rho = c.random_direction()
        rtol = 1e-13 if m < 50 else 2e-3
        assert np.allclose(rho, g["rho"][m], rtol=rtol, atol=rtol * 1e-1), m
        eta = c.fit(g["rho"][m])
        d = c.get("diag")
        assert d["x0"] == g["x0"][m] and d["init2"] == g["init"][m][2]
        assert np.array_equal(c.get("rng"), g["state_mh"][m]), m           # 1 + 3 normals consumed (sub_klhr_sinh.py:177,184)
        same = d["nfev1"] == g["nfev1"][m] and d["nfev2"] == g["nfev2"][m]
        mism += not same
        if same and g["status2"][m] == 0:
            assert np.allclose(eta[:3], g["eta"][m], rtol=2e-6, atol=2e-6), m
        c.set("draw", m)
        c.metropolis(np.concatenate([g["eta"][m], [0.0]]), g["rho"][m])
        d = c.get("diag")
        assert np.isclose(d["zp"], g["zp"][m], rtol=1e-11, atol=1e-12), m
        assert np.isclose(d["r"], g["r"][m], rtol=1e-9, atol=1e-9), m
        if abs(g["logu"][m] - min(0.0, g["r"][m])) > 1e-8:
            assert bool(d["accept"]) == bool(g["accept"][m]), m
        assert np.array_equal(c.get("rng"), g["state_after"][m]), m
        c.set("theta", g["theta_after"][m])
        c.adapt()
    assert mism <= 0.01 * M + 1, (fn, mism)


def test_sub_free_running_and_public_api(oracle):
    g = golden("g12_sub_funnel10_s0_w100.npz")
    p = golden("g12_sub_public_api.npz")
    c = oracle.Chain("funnel", {"D": 9}, seed=0, warmup=100, variant="sub")
    th = c.draw_many(80)
    assert np.abs(th[:50] - g["theta_after"][:50]).max() < 1e-9
    assert np.abs(th[:50] - p["s0_thetas"][:50]).max() < 1e-9
    assert c.get("grad_evals") > 0

"""CPU tests of the host side: the C-ABI library loads and exports every symbol include/klhr.h declares (no compute
calls without a GPU), host helpers match the reference's fixtures, host classes mirror the reference interface."""
import ctypes as C
import inspect
import os
import re

import numpy as np
import pytest

from conftest import ROOT, golden


@pytest.fixture(scope="module")
def lib():
    from klhr_amd import build as kb
    kb.build()
    from klhr_amd import _lib
    return _lib


def test_library_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "klhr.h")).read()
    declared = sorted(set(re.findall(r"KLHR_API[^;(]*?\b(klhr_\w+)\s*\(", hdr)))
    assert len(declared) >= 25
    L = lib.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert sorted(lib.EXPORTS) == declared
    assert L.klhr_abi_version() == 1


def test_config_struct_layout_matches_header(lib):
    # the header's struct must be the one ctypes mirrors: compile a tiny C program and compare sizeof/offsetof
    import subprocess
    import tempfile
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "klhr.h"
int main(void){ printf("%zu %zu %zu %zu %zu %zu\n", sizeof(klhr_config), offsetof(klhr_config,l), offsetof(klhr_config,warmup),
  offsetof(klhr_config,scale_dir_cov), offsetof(klhr_config,quad_x), offsetof(klhr_config,quad_asinh_x)); return 0; }'''
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", os.path.join(td, "t"), os.path.join(td, "t.c")])
        got = [int(x) for x in subprocess.check_output([os.path.join(td, "t")]).split()]
    Cfg = lib.Config
    want = [C.sizeof(Cfg), Cfg.l.offset, Cfg.warmup.offset, Cfg.scale_dir_cov.offset, Cfg.quad_x.offset, Cfg.quad_asinh_x.offset]
    assert got == want


def test_compute_fails_loudly_without_gpu(lib):
    if os.path.exists("/dev/kfd"):
        pytest.skip("a GPU is present")
    with pytest.raises(lib.KlhrError):
        lib.Sampler(lib.MODEL_IDS["funnel"], np.array([9.0]), 8)


def test_create_rejects_unsupported_configs(lib):
    for kw in (dict(N=4), dict(J=5), dict(overrelaxed=True, K=0)):
        with pytest.raises(lib.KlhrError):
            lib.Sampler(lib.MODEL_IDS["funnel"], np.array([9.0]), 8, **kw)


def test_host_helpers_vs_golden(lib):
    g = golden("g2_rng.npz")
    for s in g["seeds"]:
        assert np.array_equal(lib.seed_state(int(s)), g[f"s{int(s)}_state0"])
    g7 = golden("g7_online.npz")
    for i, (wu, ws, sc) in enumerate(g7["wa_settings"]):
        assert lib.window_closures(int(wu), int(ws), int(sc)) == list(g7[f"wa{i}_closures"])


def test_windowed_adaptation_class_vs_golden():
    from klhr_amd import WindowedAdaptation
    g7 = golden("g7_online.npz")
    for i, (wu, ws, sc) in enumerate(g7["wa_settings"]):
        if f"wa{i}_closed" not in g7.files:
            continue
        wa = WindowedAdaptation(int(wu), windowsize=int(ws), windowscale=int(sc))
        closed = [m for m in range(1, 1200) if wa.window_closed(m)]
        assert closed == list(g7[f"wa{i}_closed"])


def test_online_classes_vs_golden():
    from klhr_amd import OnlineMoments, OnlinePCA
    g = golden("g7_online.npz")
    for D in (3, 10):
        om, pca = OnlineMoments(D), OnlinePCA(D, K=2, l=4)
        for k, x in enumerate(g[f"D{D}_X"]):
            om.update(x)
            pca.update(x)
            assert np.array_equal(om.mean(), g[f"D{D}_m"][k]) and np.array_equal(om.v, g[f"D{D}_v"][k])
            assert np.array_equal(om.var(), g[f"D{D}_var"][k])
            assert np.allclose(pca.v[:, 0], g[f"D{D}_pcav"][k][:, 0], rtol=1e-12, atol=1e-14)
            assert np.allclose(pca.values()[0], g[f"D{D}_pcavals"][k][0], rtol=1e-12)
        om.reset()
        assert om.N == 0 and not om.m.any()
    # batched form: one object tracks many chains
    X = np.random.default_rng(0).normal(size=(20, 5, 3))
    b = OnlineMoments(3, batch=(5,))
    singles = [OnlineMoments(3) for _ in range(5)]
    for x in X:
        b.update(x)
        for c in range(5):
            singles[c].update(x[c])
    assert np.array_equal(b.var(), np.array([s.var() for s in singles]))


def test_reference_interface_is_mirrored():
    """Same class names, constructor keywords and defaults as klhr_sinh.py:15-32, mh.py:8, bsmodel.py:6-7."""
    from klhr_amd.bsmodel import BSModel
    from klhr_amd.klhr_sinh import KLHRSINH
    from klhr_amd.mh import MH
    sig = inspect.signature(KLHRSINH.__init__).parameters
    ref = dict(theta=None, seed=None, N=8, K=10, J=2, l=4, initscale=0.1, warmup=1000, windowsize=50, windowscale=2,
               tol=1e-10, grad_clip=1e15, scale_clip=300, scale_dir_cov=False, overrelaxed=True, eigen_method_one=False,
               max_init_tries=100)
    for k, v in ref.items():
        assert sig[k].default == v, k
    assert list(sig)[1] == "bsmodel"
    for m in ("draw", "sample", "fit", "KL", "logp_grad_rho", "_random_direction", "_metropolis_step", "_T", "_T_inv",
              "_log_q", "_log_abs_jac", "_grad_T", "_grad_log_abs_jac", "__iter__", "__next__", "log_density"):
        assert hasattr(KLHRSINH, m), m
    assert list(inspect.signature(MH.__init__).parameters)[1:5] == ["model", "stepsize", "seed", "theta"]
    bs = inspect.signature(BSModel.__init__).parameters
    assert bs["stan_file"].default == "" and bs["data_file"].default == "" and bs["stepsize"].default == 1.0
    for m in ("dim", "log_density", "log_density_gradient", "log_density_hessian", "Hamiltonian", "unconstrain",
              "constrain", "parameter_names"):
        assert hasattr(BSModel, m), m


def test_bsmodel_registry_and_packing():
    from klhr_amd.bsmodel import BSModel
    stan = os.path.join(ROOT, "klhr_amd", "stan")
    m = BSModel(os.path.join(stan, "glmm-poisson.stan"), os.path.join(stan, "glmm-poisson.json"))
    assert m.dim() == 45 and m.model_data.size == 81 and m.model_data[0] == 40
    th = np.arange(45) * 0.01 + 0.1
    assert np.allclose(m.unconstrain(m.constrain(th)), th)
    f = BSModel(os.path.join(stan, "funnel.stan"), os.path.join(stan, "funnel.json"))
    assert f.dim() == 10 and f.parameter_names()[0] == "double_log_sigma"
    with pytest.raises(ValueError):
        BSModel("stan/hmm.stan", os.path.join(stan, "funnel.json"))


def test_ess_estimator_on_ar1():
    from klhr_amd.ess import ess, ess_ar1_exact
    rng = np.random.default_rng(1)
    M, C, phi = 4000, 16, 0.8
    x = np.zeros((M, C))
    x[0] = rng.normal(size=C)
    for t in range(1, M):
        x[t] = phi * x[t - 1] + np.sqrt(1 - phi * phi) * rng.normal(size=C)
    e = ess(x)
    want = C * ess_ar1_exact(M, phi)
    assert 0.8 * want < e < 1.25 * want, (e, want)
    iid = rng.normal(size=(2000, 8))
    assert 0.8 * 16000 < ess(iid) < 1.3 * 16000


def test_ess_detail_reports_tau_lag_and_a_saturated_window():
    """ess_detail: tau and the truncation lag of a resolved chain; a chain whose autocorrelation time exceeds the run leaves the
    truncation at the end of the window (what bench.py refuses to quote); the between-chain tau bench.py cross-checks with."""
    from klhr_amd.ess import ess_detail
    rng = np.random.default_rng(3)
    M, C, phi = 20000, 64, 0.9
    x = np.zeros((M, C))
    x[0] = rng.normal(size=C) / np.sqrt(1 - phi * phi)
    e = rng.normal(size=(M, C))
    for t in range(1, M):
        x[t] = phi * x[t - 1] + e[t]
    d = ess_detail(x)
    tau_exact = (1 + phi) / (1 - phi)
    assert 0.85 * tau_exact < d["tau"] < 1.15 * tau_exact and d["truncation_lag"] < M / 10 and abs(d["rhat"] - 1) < 0.01
    tau_between = M * x.mean(axis=0).var(ddof=1) / x.var()
    assert 0.6 * tau_exact < tau_between < 1.5 * tau_exact                 # (64 chains: +-18 %)
    slow = np.cumsum(rng.normal(size=(500, 8)), axis=0)                    # random walks: tau >= run length
    ds = ess_detail(slow)
    assert ds["truncation_lag"] >= 500 / 10


def test_pooled_diag_is_the_diagonal_of_pooled_full():
    from klhr_amd.parallel import pooled_diag, pooled_full
    rng = np.random.default_rng(5)
    X = rng.normal(size=(300, 6)) * np.arange(1, 7)
    stats = np.concatenate([[X.shape[0]], X.sum(axis=0), (X.T @ X).reshape(-1)])
    m, v = pooled_diag(stats, 6)
    fm, fd, _, _ = pooled_full(stats, 6, 2)
    assert np.allclose(m, X.mean(axis=0), rtol=1e-13) and np.allclose(v, X.var(axis=0, ddof=1), rtol=1e-11)
    assert np.allclose(m, fm, rtol=0, atol=0) and np.allclose(v, fd, rtol=1e-12)


def test_traffic_entries_are_tied_to_a_build():
    """every round-3 entry of profiles/traffic.json names the library build it was measured on (bench.py quotes it for that build only)"""
    import json
    from klhr_amd import build
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    keys = ["funnel-10-65536", "ill-normal-100-16384", "glmm-poisson-45-8192", "funnel-100-131072"]
    assert all(len(t[k].get("build_id", "")) == 64 for k in keys)
    assert len(build.source_hash()) == 64


def test_roofline_byte_model():
    import bench
    assert bench.algorithmic_bytes_per_iter(10) == 1328
    assert bench.algorithmic_bytes_per_iter(45) == 5528
    assert bench.algorithmic_bytes_per_iter(100) == 12128


def test_window_closures_degenerate_schedules(lib):
    """klhr_window_closures on the inputs where a naive restatement of WindowedAdaptation._calculate_windows
    (windowedadaptation.py:12-28) does not terminate or overflows: windowsize == 0, windowscale <= 0, windowscale == 1 with
    many windows.  Expected values: the reference class itself, run in the build container (first 12 closures, count, last)."""
    expected = {(100, 0, 2): ([0], 1, 0), (100, 50, 0): ([50], 1, 50), (100, 50, -1): ([50, 100], 2, 100), (100, 50, 1): ([50, 100], 2, 100),
                (1000, 50, 1): ([50, 100, 150, 200, 250, 300, 350, 400, 450, 500, 550, 600], 20, 1000), (100, -5, 2): ([], 0, None),
                (10, 3, 3): ([3, 10], 2, 10), (200, 7, 1): ([7, 14, 21, 28, 35, 42, 49, 56, 63, 70, 77, 84], 29, 200), (0, 50, 2): ([], 0, None),
                (50, 50, 2): ([], 0, None), (51, 50, 2): ([50, 51], 2, 51),
                (5000000, 50, 2): ([50, 150, 350, 750, 1550, 3150, 6350, 12750, 25550, 51150, 102350, 204750], 17, 5000000),
                (100, 1, 1): (list(range(1, 13)), 100, 100)}
    for (wu, ws, sc), (first, n, last) in expected.items():
        got = lib.window_closures(wu, ws, sc)
        assert len(got) == n and got[:12] == first and (n == 0 or got[-1] == last), (wu, ws, sc, got[:14], len(got))
    # extreme arguments terminate too (128-bit intermediates): no hang, no overflow
    assert lib.window_closures(2**62, 1, 2**31) == [1, 2**62]        # 1 + 2^31 * 2^31 >= warmup: the second window runs to warmup
    assert len(lib.window_closures(2**62, 2**40, 3)) < 64


def test_pcg64_jump_constants_of_the_quad_generator():
    """device_rng.hpp: rng_std_normal_quad advances PCG64's 128-bit LCG q + 1 and 4 steps at a time: state * M^k + inc * G_k with
    G_k = 1 + M + ... + M^(k-1).  The constants in the header must be those powers / sums of NumPy's multiplier (mod 2^128), and
    k single steps must equal one jump for an arbitrary state and increment."""
    import re
    src = open(os.path.join(ROOT, "klhr_amd", "csrc", "device_rng.hpp")).read()
    body = src[src.index("rng_std_normal_quad(Pcg64 &g"):]
    val = {}
    for name, hi, lo in re.findall(r"\b([MG][1-4]) = \(\(\(u128\)0x([0-9A-Fa-f]+)ULL\) << 64\) \| \(u128\)0x([0-9A-Fa-f]+)ULL", body):
        val[name] = (int(hi, 16) << 64) | int(lo, 16)
    assert re.search(r"\bG1 = 1\b", body)
    val["G1"] = 1
    assert sorted(val) == ["G1", "G2", "G3", "G4", "M1", "M2", "M3", "M4"], sorted(val)
    mod = 1 << 128
    M = (0x2360ED051FC65DA4 << 64) | 0x4385DF649FCCF645          # numpy/random/src/pcg64/pcg64.h: PCG_DEFAULT_MULTIPLIER
    assert val["M1"] == M
    g = 0
    for k in range(1, 5):
        g = (g * M + 1) % mod
        assert val[f"M{k}"] == pow(M, k, mod), k
        assert val[f"G{k}"] == g, k
    state, inc = 0x0123456789ABCDEF_FEDCBA9876543210, (0x9E3779B97F4A7C15_F39CC0605CEDC835 | 1)
    t = state
    for k in range(1, 5):
        t = (t * M + inc) % mod
        assert t == (state * val[f"M{k}"] + inc * val[f"G{k}"]) % mod

"""Host-logic test of the PRODUCT's optimiser state machine (klhr_amd/csrc/device_opt.hpp): compiled for the CPU
(oracle/opt_harness.cpp) and fed the oracle's objective functions, it must reproduce the oracle's straight-line
restatement of SciPy's BFGS/DCSRCH/wolfe2 exactly — same x, same evaluation count, same status — because the two are
then running identical arithmetic.  This isolates control-flow bugs from GPU floating-point differences."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, golden, model_spec

HARNESS = os.path.join(ROOT, "oracle", "libklhr_harness.so")


@pytest.fixture(scope="module")
def harness(oracle):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libklhr_harness.so"], stdout=subprocess.DEVNULL)
    H = C.CDLL(HARNESS)
    dp = C.POINTER(C.c_double)
    H.klhr_harness_bfgs.argtypes = [C.c_void_p, dp, C.c_int, dp, C.c_double, dp]
    L = oracle.lib()
    L.klhr_oracle_bfgs.argtypes = [C.c_void_p, dp, C.c_int, dp, C.c_double, dp]
    return H, L


def run_both(harness, chain, rho, n, x0, gtol):
    H, L = harness
    dp = C.POINTER(C.c_double)
    rho = np.ascontiguousarray(rho)
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    a, b = np.zeros(10), np.zeros(10)
    L.klhr_oracle_bfgs(chain._h, rho.ctypes.data_as(dp), n, x0.ctypes.data_as(dp), gtol, a.ctypes.data_as(dp))
    H.klhr_harness_bfgs(chain._h, rho.ctypes.data_as(dp), n, x0.ctypes.data_as(dp), gtol, b.ctypes.data_as(dp))
    return a, b


@pytest.mark.parametrize("fn,name,nrec", [("g8_funnel10_s0_w100.npz", "funnel10", 1000),
                                          ("g8_funnel10_s7_w1000.npz", "funnel10", 400),
                                          ("g8_glmm45_s0_w100.npz", "glmm45", 160),
                                          ("g8_glmm45_s1_w100.npz", "glmm45", 160),
                                          ("g8_illnormal100_s0_w100.npz", "illnormal100", 160),
                                          ("g8_normal2_s0_w100.npz", "normal2", 300)])
def test_state_machine_equals_straight_line_bfgs(oracle, harness, fn, name, nrec):
    g = golden(fn)
    stem, data = model_spec(name)
    c = oracle.Chain(stem, data, seed=0, warmup=100)
    n_fallback = 0
    for m in range(nrec):
        c.set("theta", g["theta_before"][m])
        rho = g["rho"][m]
        a, b = run_both(harness, c, rho, 1, np.array([g["x0"][m]]), 1e-5)
        assert np.array_equal(a, b, equal_nan=True), (m, "stage1", a, b)
        a, b = run_both(harness, c, rho, 4, g["init"][m], 1e-3)
        assert np.array_equal(a, b, equal_nan=True), (m, "stage2", a, b)
        n_fallback += int(a[9] > 0)
    if name == "glmm45":
        assert n_fallback > 100      # the wolfe2 / zoom fallback path is really exercised


def test_state_machine_pathological_starts(oracle, harness):
    """Bad starting points: huge steps, flat directions, NaN/inf objectives."""
    stem, data = model_spec("funnel10")
    c = oracle.Chain(stem, data, seed=0, warmup=100)
    rng = np.random.default_rng(3)
    for k in range(300):
        th = rng.normal(size=10) * (0.3 if k % 3 else 3.0)
        if k % 7 == 0:
            th[0] = -12.0
        if k % 11 == 0:
            th[0] = 9.0
        c.set("theta", th)
        rho = rng.normal(size=10)
        rho /= np.linalg.norm(rho)
        x0 = rng.normal(size=4) * (0.1 if k % 5 else 4.0)
        if k % 13 == 0:
            x0[1] = 40.0
        a, b = run_both(harness, c, rho, 4, x0, 1e-3)
        assert np.array_equal(a, b, equal_nan=True), (k, a, b)
        a, b = run_both(harness, c, rho, 1, x0[:1] * 10, 1e-5)
        assert np.array_equal(a, b, equal_nan=True), (k, a, b)


def test_rank_two_update_agrees_with_scipy_form(oracle, harness):
    """The GPU build forms the inverse-Hessian update in its rank-two expansion (device_opt.hpp, KLHR_BFGS_RANK2); the
    host harness compiled the same way must follow the oracle's SciPy-literal iteration to rounding: identical
    evaluation counts in >= 97 % of the golden funnel records, and the same minimiser there."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libklhr_harness_r2.so"], stdout=subprocess.DEVNULL)
    H2 = C.CDLL(os.path.join(ROOT, "oracle", "libklhr_harness_r2.so"))
    dp = C.POINTER(C.c_double)
    H2.klhr_harness_bfgs.argtypes = [C.c_void_p, dp, C.c_int, dp, C.c_double, dp]
    g = golden("g8_funnel10_s0_w100.npz")
    stem, data = model_spec("funnel10")
    c = oracle.Chain(stem, data, seed=0, warmup=100)
    same, worst = 0, 0.0
    N = 600
    for m in range(N):
        c.set("theta", g["theta_before"][m])
        a, b = run_both((H2, harness[1]), c, g["rho"][m], 4, g["init"][m], 1e-3)
        if a[6] == b[6] and a[8] == b[8]:
            same += 1
            worst = max(worst, float(np.max(np.abs(a[:4] - b[:4]) / (1 + np.abs(a[:4])))))
    assert same >= 0.97 * N, same
    assert worst < 1e-8, worst


def test_dimension_specialised_optimiser_equals_padded_form(oracle, harness):
    """opt_advance<OUTLINE, N> (round 2: the mode search runs N = 1, the KL fits N = 4 | 3 | 2) against the zero-padded
    N = 4 form every problem ran through in round 1: identical x, value, Hessian entry, counts and status, bit for bit, on the
    golden funnel records and on pathological starts (padded coordinates only ever added exact zeros)."""
    H, _ = harness
    dp = C.POINTER(C.c_double)
    H.klhr_harness_bfgs_n.argtypes = [C.c_void_p, dp, C.c_int, dp, C.c_double, dp]
    g = golden("g8_funnel10_s0_w100.npz")
    stem, data = model_spec("funnel10")
    c = oracle.Chain(stem, data, seed=0, warmup=100)
    rng = np.random.default_rng(11)
    for m in range(400):
        c.set("theta", g["theta_before"][m] if m % 4 else rng.normal(size=10) * 2.5)
        rho = np.ascontiguousarray(g["rho"][m])
        for n, x0, gtol in ((1, np.array([g["x0"][m] * (1 if m % 5 else 30)]), 1e-5), (4, np.ascontiguousarray(g["init"][m]), 1e-3)):
            a, b = np.zeros(10), np.zeros(10)
            H.klhr_harness_bfgs(c._h, rho.ctypes.data_as(dp), n, x0.ctypes.data_as(dp), gtol, a.ctypes.data_as(dp))
            H.klhr_harness_bfgs_n(c._h, rho.ctypes.data_as(dp), n, x0.ctypes.data_as(dp), gtol, b.ctypes.data_as(dp))
            assert np.array_equal(a, b, equal_nan=True), (m, n, a, b)

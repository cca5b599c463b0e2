"""GPU parity tests proper: the HIP path (through the C ABI, klhr_amd/_lib.py) against the CPU oracle on the same
seeded inputs and against the committed golden fixtures.  Run on the MI355X box: pytest -m gpu."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden, model_spec
from parity_util import STATE_FIELDS, oracle_state_trace, make_sampler, oracle_teacher_forced, parse_g8

pytestmark = pytest.mark.gpu

MODELS = ["funnel10", "illnormal100", "glmm45", "normal2", "funnel100", "ar1x100", "earnings"]


def relerr(a, b, floor=1.0):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b) / np.maximum(np.abs(b), floor)


def test_seedsequence_on_host_helper(oracle):
    from klhr_amd import _lib
    for s in [0, 1, 12345, 2**32 + 5, 2**63 + 11]:
        assert np.array_equal(_lib.seed_state(s), oracle.seed_state(s))


@pytest.mark.parametrize("name", MODELS)
def test_seed_and_initialize_bit_exact(oracle, name):
    """MCMCBase.__init__ + _initialize on device: theta0 and the RNG position equal the oracle's, bit for bit."""
    stem, data = model_spec(name)
    C = 64
    s = make_sampler(stem, data, C, warmup=100)
    seeds = np.arange(C, dtype=np.uint64) + 7
    s.seed(seeds)
    th, rng = s.get("theta"), s.get("rng")
    for c in range(C):
        o = oracle.Chain(stem, data, seed=int(seeds[c]), warmup=100)
        assert np.array_equal(o.get("rng"), rng[c]), c
        assert np.array_equal(o.get("theta"), th[c]), c
    if name == "funnel10":
        g = golden("g8_funnel10_s0_w100.npz")
        s0 = make_sampler(stem, data, 1, warmup=100)
        s0.seed(np.array([0], dtype=np.uint64))
        assert np.array_equal(s0.get("theta")[0], g["theta0"])
        assert np.array_equal(s0.get("rng")[0], g["state_before"][0])


@pytest.mark.parametrize("name", MODELS)
def test_models_kl_line_vs_golden_and_oracle(oracle, name):
    """a7/a8/a9 + a-L: log density, gradient, KL value/gradient, line objective on the G4 inputs."""
    g = golden("g4_kl.npz")
    stem, data = model_spec(name)
    th, rho, eta = g[f"{name}_theta"], g[f"{name}_rho"], g[f"{name}_eta"]
    n = th.shape[0]
    s = make_sampler(stem, data, n, warmup=100)
    lp, grad = s.log_density_gradient(th)
    fin = np.isfinite(g[f"{name}_lp"])
    assert np.all(relerr(lp[fin], g[f"{name}_lp"][fin]) < 1e-12)
    assert np.all(relerr(grad[fin], g[f"{name}_lpgrad"][fin]) < 1e-12)
    lp_only, _ = s.log_density_gradient(th, grad=False)
    assert np.array_equal(lp_only, lp)
    s.set("theta", th)
    v, gr = s.kl_eval(eta, rho)
    ref_v, ref_g = g[f"{name}_kl"], g[f"{name}_klgrad"]
    ok = np.isfinite(ref_v) & np.all(np.isfinite(ref_g), axis=1)
    assert ok.sum() >= n // 2
    scale = np.maximum(1.0, np.abs(ref_v[ok]))
    assert np.all(np.abs(v[ok] - ref_v[ok]) <= 1e-12 * scale), np.max(np.abs(v[ok] - ref_v[ok]) / scale)
    gs = np.maximum(1.0, np.abs(ref_g[ok]).max(axis=1))[:, None]
    assert np.all(np.abs(gr[ok] - ref_g[ok]) <= 1e-11 * gs), np.max(np.abs(gr[ok] - ref_g[ok]) / gs)
    # non-finite cases must be non-finite in the same way
    bad = ~ok
    assert np.all(~np.isfinite(v[bad]) | ~np.all(np.isfinite(gr[bad]), axis=1))
    line = s.line_eval(eta[:, 0], rho)
    ref_l = g[f"{name}_line"]
    okl = np.all(np.isfinite(ref_l), axis=1)
    assert np.all(np.abs(line[okl] - ref_l[okl]) <= 1e-12 * np.maximum(1.0, np.abs(ref_l[okl]).max(axis=1))[:, None])
    # and against the oracle directly
    for i in range(0, n, 7):
        o = oracle.Chain(stem, data, seed=1)
        o.set("theta", th[i])
        vo, go = o.KL(eta[i], rho[i])
        if np.isfinite(vo):
            assert abs(v[i] - vo) <= 1e-12 * max(1.0, abs(vo))


@pytest.mark.parametrize("name", MODELS)
def test_kl_with_the_gradient_clip_active_vs_reference(oracle, name):
    """G4c (the reference's KL with scale_clip = 3: the clip bites at nearly every node, ar1 included).  On the closed-form
    kernels every such node fails line_point's clip bound and is recomputed by the exact element-wise patch (lane kernel:
    model_eval_nv<MODEL, 1> over the LDS row; split pipeline: exact_points4 from the tiled rows) — checked here against the
    reference's numbers, in every kernel mode, and against the oracle."""
    g = golden("g4c_kl_clip.npz")
    if f"{name}_theta" not in g.files:
        pytest.skip("no G4c points for this target")
    stem, data = model_spec(name)
    th, rho, eta = g[f"{name}_theta"], g[f"{name}_rho"], g[f"{name}_eta"]
    n = th.shape[0]
    sc = float(g["scale_clip"][0])
    s = make_sampler(stem, data, n, warmup=100, scale_clip=sc)
    s.set("theta", th)
    v, gr = s.kl_eval(eta, rho)
    ref_v, ref_g = g[f"{name}_kl"], g[f"{name}_klgrad"]
    ok = np.isfinite(ref_v) & np.all(np.isfinite(ref_g), axis=1)
    assert ok.sum() >= n - 2 and (g[f"{name}_nclip"][ok] > 0).sum() >= 58
    scale = np.maximum(1.0, np.abs(ref_v[ok]))
    assert np.all(np.abs(v[ok] - ref_v[ok]) <= 1e-12 * scale), np.max(np.abs(v[ok] - ref_v[ok]) / scale)
    gs = np.maximum(1.0, np.abs(ref_g[ok]).max(axis=1))[:, None]
    assert np.all(np.abs(gr[ok] - ref_g[ok]) <= 1e-11 * gs), np.max(np.abs(gr[ok] - ref_g[ok]) / gs)
    for i in range(0, n, 5):
        o = oracle.Chain(stem, data, seed=1, scale_clip=sc)
        o.set("theta", th[i])
        vo, go = o.KL(eta[i], rho[i])
        if np.isfinite(vo):
            assert abs(v[i] - vo) <= 1e-12 * max(1.0, abs(vo))


def test_primitives_vs_golden():
    g = golden("g3_primitives.npz")
    stem, data = model_spec("normal2")
    s = make_sampler(stem, data, 8)
    etas, xs, ys = g["etas"], g["xs"], g["ys"]
    E = np.repeat(etas, xs.size, axis=0)
    X = np.tile(xs, etas.shape[0])
    out = s.primitives(X, E).reshape(etas.shape[0], xs.size, 12)

    def close(a, b, rtol, atol):
        both_nan = np.isnan(a) & np.isnan(b)
        same_inf = np.isinf(a) & np.isinf(b) & (np.sign(a) == np.sign(b))
        fin = np.isfinite(a) & np.isfinite(b)
        with np.errstate(invalid="ignore"):
            return np.all(both_nan | same_inf | (fin & (np.abs(a - b) <= atol + rtol * np.abs(b))))
    assert close(out[..., 0], g["T"], 1e-12, 1e-13)
    assert close(out[..., 2], g["log_abs_jac"], 1e-12, 1e-12)
    assert close(out[..., 4:8], g["grad_T"], 1e-12, 1e-13)
    assert close(out[..., 8:12], g["grad_log_abs_jac"], 1e-12, 1e-13)
    E = np.repeat(etas, ys.size, axis=0)
    Y = np.tile(ys, etas.shape[0])
    out = s.primitives(Y, E).reshape(etas.shape[0], ys.size, 12)
    assert close(out[..., 1], g["T_inv"], 1e-11, 1e-13)
    assert close(out[..., 3], g["log_q"], 1e-11, 1e-11)


G8 = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "g8_*_w*.npz")))


def ill_conditioned(oracle, R, m, init):
    """How far the ORACLE's own KL fit of record m moves (relative, max over eta) when one component of init changes by one
    ulp (each of the four, either way; the largest movement counts): the conditioning of that record's optimisation problem.
    ~1e-14 for an ordinary record, > 1e-4 for the rare fit that walks dozens of evaluations along a flat valley — there no
    two implementations (not SciPy's own two algebraically equal forms of the BFGS update either) end at the same point."""
    import ctypes as C
    L = oracle.lib()
    dp = C.POINTER(C.c_double)
    L.klhr_oracle_bfgs.argtypes = [C.c_void_p, dp, C.c_int, dp, C.c_double, dp]
    c = oracle.Chain(R["stem"], R["data"], seed=0, warmup=R["warmup"], **R["flags"])
    c.set("theta", R["g"]["theta_before"][m])
    rho = np.ascontiguousarray(R["g"]["rho"][m])
    def run(x0):
        out = np.zeros(10)
        L.klhr_oracle_bfgs(c._h, rho.ctypes.data_as(dp), 4, x0.ctypes.data_as(dp), 1e-3, out.ctypes.data_as(dp))
        return out[:4].copy()
    base = run(np.array(init, dtype=np.float64))
    worst = 0.0
    for k in range(4):
        for towards in (10.0, -10.0):
            x0 = np.array(init, dtype=np.float64)
            x0[k] = np.nextafter(x0[k], towards)
            worst = max(worst, float(np.max(np.abs(run(x0) - base) / (1.0 + np.abs(base)))))
    return worst


@pytest.mark.parametrize("fn", G8)
def test_teacher_forced_pieces_vs_oracle(oracle, fn):
    """Every record of a golden chain trace becomes one GPU chain: same state before the draw, then each piece of
    draw() is run on the device and compared with the oracle's result for that record (and with the reference's)."""
    # (the four cfg1 fixtures — funnel-10, seeds 0-3, 1 000 draws each — are teacher-forced over all their records, the others over 400)
    R = oracle_teacher_forced(oracle, fn, max_records=None if fn.startswith("g8_funnel10_s") and fn[13] in "0123" and fn.endswith("_w100.npz") else 400)
    g, M, flags = R["g"], R["M"], R["flags"]
    s = make_sampler(R["stem"], R["data"], M, warmup=R["warmup"], **flags)
    s.set("theta", g["theta_before"][:M])
    s.set("rng", g["state_before"][:M])
    for f in R["fields"]:
        s.set(f, R["before"][f])
    s.set("counters", R["cnt_before"])
    # ---- a5 direction: designed to be bit-exact given identical adaptation state
    rho = s.random_direction()
    assert np.array_equal(s.get("rng"), R["rng_after_dir"])
    assert np.array_equal(rho, R["rho"]), np.abs(rho - R["rho"]).max()
    # ---- a10 fit from the reference's rho
    s.set("grad_evals", np.asarray(R["gev_before"], dtype=np.int64))
    eta = s.fit(g["rho"][:M])
    assert np.array_equal(s.get("rng"), R["rng_after_fit"])
    d = s.get("diag")
    from klhr_amd._lib import DIAG_COLS as DC
    col = {k: i for i, k in enumerate(DC)}
    o_diag = R["diag_fit"]
    o_nfev1 = np.array([x["nfev1"] for x in o_diag]); o_nfev2 = np.array([x["nfev2"] for x in o_diag])
    o_st1 = np.array([x["status1"] for x in o_diag]); o_st2 = np.array([x["status2"] for x in o_diag])
    o_x0 = np.array([x["x0"] for x in o_diag])
    assert np.array_equal(d[:, col["x0"]], o_x0)                                   # RNG-derived start point: exact
    assert np.array_equal(d[:, col["init2"]], np.array([x["init2"] for x in o_diag]))
    assert np.array_equal(d[:, col["init3"]], np.array([x["init3"] for x in o_diag]))
    same = (d[:, col["nfev1"]] == o_nfev1) & (d[:, col["nfev2"]] == o_nfev2)
    conv = (o_st1 == 0) & (o_st2 == 0)
    # converged fits: identical evaluation counts in >= 99 % of records, eta within 2e-6 there
    # (ar1 is Gaussian, so the KL objective is flat in the tail parameters; earnings has a 1e4:1 scale spread between
    #  coordinates: on both more fits sit on a Wolfe / termination decision boundary and stop one evaluation apart —
    #  the same spread the reference shows against the CPU oracle in tests/test_oracle_chain.py.  Those records must
    #  still reach the same KL value.)
    need = {"ar1": 0.95, "earnings": 0.95}.get(R["stem"], 0.99)
    if conv.sum() > 20:
        assert (same & conv).sum() >= need * conv.sum(), ((same & conv).sum(), conv.sum())
    off = conv & ~same
    if off.any():
        o_f2 = np.array([x["fun2"] for x in o_diag])
        # (both stopped on gtol = 1e-3 — the KL values agree to the level that tolerance implies, not to rounding)
        assert np.all(np.abs(d[off, col["fun2"]] - o_f2[off]) <= 2e-3 * (1.0 + np.abs(o_f2[off]))), \
            np.abs(d[off, col["fun2"]] - o_f2[off]).max()
    sel = same & conv
    err = np.abs(eta - R["eta"]) / (1.0 + np.abs(R["eta"]))
    # Converged fits with identical evaluation counts: eta agrees at the level the kernels deliver (tools/parity_stats.py,
    # round 2, lane / octet / split kernels alike: worst record 1e-11 on funnel / normal / ill-normal / funnel-100,
    # 9e-11 on ar1, 3e-9 on glmm-poisson, 2e-7 on earnings whose +-300 clip leaves fits at rounding-noise level), with one
    # decade of margin.  A record may exceed it only if the ORACLE ITSELF is that sensitive there: its own minimiser moves
    # by more than 1e-7 when a component of init is changed by one ulp (ill_conditioned below).  g8_funnel10_s7_w1000 holds one such
    # record (#137: a 34-40 evaluation walk along a flat valley of KL; the oracle lands 2.3e-3 away after a one-ulp nudge,
    # SciPy-literal and rank-two forms of the same BFGS update 1.0e-3 apart, the reference itself 8e-4 from the oracle).
    tol_eta = {"earnings": 2e-6, "glmm-poisson": 1e-7}.get(R["stem"], 1e-9)
    worst_o = err[sel].max(axis=1) if sel.any() else np.zeros(1)
    over = np.flatnonzero(sel)[worst_o > tol_eta] if sel.any() else np.zeros(0, dtype=int)
    assert over.size <= max(1, int(0.01 * sel.sum())), (over, worst_o[worst_o > tol_eta])
    for m in over:
        o_rec = o_diag[m]
        init = [o_rec["init0"], o_rec["init1"], o_rec["init2"], o_rec["init3"]]
        moved = ill_conditioned(oracle, R, m, init)
        print(f"{fn}: record {m}: eta differs by {worst_o[np.flatnonzero(sel) == m][0]:.2e} at equal evaluation counts; "
              f"the oracle's own minimiser moves by {moved:.2e} under a one-ulp change of one component of init")
        assert moved > 1e-7, (fn, m, moved)
        # the allowance is tied to the measured sensitivity: the device may be off by at most 100 x what a one-ulp nudge does to the oracle
        assert worst_o[np.flatnonzero(sel) == m][0] <= 100.0 * moved, (fn, m, worst_o[np.flatnonzero(sel) == m][0], moved)
        # ... and both fits reached the same objective value to the level gtol = 1e-3 implies
        assert abs(d[m, col["fun2"]] - o_rec["fun2"]) <= 1e-5 * (1.0 + abs(o_rec["fun2"])), (fn, m)
    assert np.all(d[sel, col["status1"]] == 0) and np.all(d[sel, col["status2"]] == 0)
    # Fits that end in a failed line search (status 2; the norm on glmm-poisson, where the +-300 gradient clip makes
    # the gradient inconsistent with the objective) stop at rounding-noise level: WHICH iterate is the last one is
    # decided by the last bits of f, so x may differ along flat directions while the achieved KL value agrees.
    fl = ~conv
    if fl.any():
        o_fun2 = np.array([x["fun2"] for x in o_diag])
        with np.errstate(invalid="ignore"):
            frel = np.abs(d[fl, col["fun2"]] - o_fun2[fl]) / (1.0 + np.abs(o_fun2[fl]))
        # (wolfe2 can accept a step whose f is NaN when only the derivative test holds — SciPy then stops with
        #  status 2 and fun = NaN; whether that happens is again decided by rounding, so allow it in <= 2 % of records)
        assert np.mean(np.nan_to_num(frel, nan=1.0) < 1e-7) >= 0.98, np.sort(frel)[-5:]
        assert np.mean(err[fl].max(axis=1) < 1e-5) >= 0.95, np.sort(err[fl].max(axis=1))[-5:]
    # grad_evals bookkeeping (klhr_sinh.py:188,200)
    ge = s.get("grad_evals")
    assert np.array_equal(ge - np.asarray(R["gev_before"], dtype=np.int64), (d[:, col["nfev1"]] + 8 * d[:, col["nfev2"]]).astype(np.int64))
    # against the reference itself
    errr = np.abs(eta - g["eta"][:M]) / (1.0 + np.abs(g["eta"][:M]))
    ref_conv = (g["status1"][:M] == 0) & (g["status2"][:M] == 0) & (d[:, col["nfev1"]] == g["nfev1"][:M]) & \
        (d[:, col["nfev2"]] == g["nfev2"][:M]) & (d[:, col["status1"]] == 0) & (d[:, col["status2"]] == 0)
    # (a record whose BFGS sits on a Wolfe/termination decision boundary may stop one step apart: still converged to
    #  gtol = 1e-3, but eta then differs at the 1e-4 level — allowed for < 1 % of records)
    worst = errr[ref_conv].max(axis=1)
    # (the reference's trajectory differs from the oracle's by OpenBLAS / glibc roundings: 5e-6 is the level
    #  tests/test_oracle_chain.py pins oracle vs reference at; beyond it only records the sensitivity check explains)
    assert np.mean(worst < 5e-6) >= 0.99, np.sort(worst)[-3:]
    # the records beyond that level: the device is there because the ORACLE is (its distance to the reference is what
    # tests/test_oracle_chain.py accounts for) — the device stays within rounding of the oracle, or the record is one of the
    # ill-conditioned ones classified above
    gap = (np.abs(R["eta"] - g["eta"][:M]) / (1.0 + np.abs(g["eta"][:M]))).max(axis=1)
    for m in np.flatnonzero(ref_conv)[worst >= 5e-6]:
        o_rec = o_diag[m]
        if not same[m]:
            continue          # stopped an evaluation apart from the oracle: a decision-boundary record, bounded through its KL value above
        near_oracle = err[m].max() <= tol_eta and gap[m] >= 0.5 * errr[m].max()
        if not near_oracle:
            moved = ill_conditioned(oracle, R, m, [o_rec["init0"], o_rec["init1"], o_rec["init2"], o_rec["init3"]])
            assert moved > 1e-7 and err[m].max() <= 100.0 * moved, (fn, m, errr[m].max(), gap[m], err[m].max(), moved)
    # ---- a12 MH step from the reference's eta and rho
    s.set("counters", R["cnt_before"])
    s.set("acc", R["acc_before"])
    s.metropolis(g["eta"][:M], g["rho"][:M])
    d = s.get("diag")
    assert np.array_equal(s.get("rng"), R["rng_after_mh"])
    o_zp = np.array([x["zp"] for x in R["diag_mh"]]); o_r = np.array([x["r"] for x in R["diag_mh"]])
    o_logu = np.array([x["logu"] for x in R["diag_mh"]]); o_acc = np.array([x["accept"] for x in R["diag_mh"]])
    assert np.all(np.abs(d[:, col["zp"]] - o_zp) <= 1e-11 * (1 + np.abs(o_zp)))
    fin = np.isfinite(o_r)
    assert np.all(np.abs(d[fin, col["r"]] - o_r[fin]) <= 1e-9 * (1 + np.abs(o_r[fin])))
    assert np.all(np.abs(d[:, col["logu"]] - o_logu) <= 1e-15 * np.abs(o_logu) + 1e-300)
    clear = np.abs(o_logu - np.minimum(0.0, np.nan_to_num(o_r, nan=-np.inf))) > 1e-8
    assert np.array_equal(d[clear, col["accept"]], o_acc[clear])
    th = s.get("theta")
    assert np.all(np.abs(th[clear] - R["theta_mh"][clear]) <= 1e-11 * (1 + np.abs(R["theta_mh"][clear])))
    assert np.allclose(s.get("acc")[clear], np.asarray(R["acc_after"])[clear], rtol=0, atol=1e-15)
    # ---- a14-a16 closure-or-update from the reference's theta: bit-exact by construction
    s.set("theta", g["theta_after"][:M])
    cb = R["cnt_before"].copy()
    cb[:, 0] += 1                      # _draw was incremented by the MH piece
    s.set("counters", cb)
    s.adapt()
    assert np.array_equal(s.get("counters"), R["cnt_after"])
    for f in R["fields"]:
        got, want = s.get(f), R["after"][f]
        if f.startswith("omd"):      # moments of the model gradient: device libm (exp) differs from glibc by ulps
            assert np.allclose(got, want, rtol=1e-12, atol=1e-13), (f, np.abs(got - want).max())
        else:
            assert np.array_equal(got, want), (f, np.abs(got - want).max())


@pytest.mark.parametrize("name", ["funnel10", "normal2"])
def test_free_running_vs_oracle_and_reference(oracle, name):
    """Chains run freely on the device (klhr_step) from the reference's seeds: first 50 draws within 1e-9 of the
    reference, accept sequence identical until the (chaotic) divergence point."""
    files = [f for f in G8 if f.startswith(f"g8_{name}_s") and f.endswith("_w100.npz")]
    stem, data = model_spec(name)
    seeds = [int(golden(f)["seed"][0]) for f in files]
    s = make_sampler(stem, data, len(files), warmup=100)
    s.seed(np.array(seeds, dtype=np.uint64))
    n = 150
    s.step(n, record_draws=True)
    dr = s.draws(0, n)
    for k, f in enumerate(files):
        g = golden(f)
        err = np.abs(dr[:, k] - g["theta_after"][:n]).max(axis=1)
        assert err[:50].max() < 1e-9, (f, err[:50].max())
        bad = np.flatnonzero(err > 1e-6)
        first = int(bad[0]) if bad.size else n
        acc_g = np.any(dr[1:first, k] != dr[:first - 1, k], axis=1)
        assert np.array_equal(acc_g, g["accept"][1:first])
    # counters after the run
    cnt = s.get("counters")
    assert np.all(cnt[:, 0] == n)


def test_chunked_launches_equal_single_launch(oracle):
    """State persists in HBM between launches: 3 + 70 + 1 iterations == 74 iterations in one call, bit for bit."""
    stem, data = model_spec("funnel10")
    seeds = np.arange(16, dtype=np.uint64) + 100
    a = make_sampler(stem, data, 16, warmup=100)
    b = make_sampler(stem, data, 16, warmup=100)
    a.seed(seeds); b.seed(seeds)
    a.step(74)
    for k in (3, 70, 1):
        b.step(k)
    for f in ["theta", "rng", "mean", "cov", "eigvecs", "eigvals", "om_m", "om_v", "pca_v", "counters", "acc", "grad_evals"]:
        assert np.array_equal(a.get(f), b.get(f)), f


def test_iterations_per_launch_and_reserved_buffer_do_not_change_results(monkeypatch):
    """KLHR_CHUNK (iterations per kernel launch) and klhr_reserve_draws only change how the work is issued: recorded
    draws and final state are bit-identical for 7, 50 and the default number of iterations per launch."""
    stem, data = model_spec("funnel10")
    seeds = np.arange(24, dtype=np.uint64) + 300
    out = []
    for chunk in (None, "7", "50"):
        if chunk is None:
            monkeypatch.delenv("KLHR_CHUNK", raising=False)
        else:
            monkeypatch.setenv("KLHR_CHUNK", chunk)
        s = make_sampler(stem, data, 24, warmup=100)
        s.seed(seeds)
        if chunk == "7":
            s.reserve_draws(120)
        s.step(120, record_draws=True)
        out.append((s.draws(0, 120), s.get("theta"), s.get("rng"), s.get("cov"), s.timing()[1]))
    if "tile" in s.kernel_name():
        assert out[0][4] == out[1][4] == out[2][4] == 1                    # whole iterations per tile: up to 256 draws per launch
    elif "stream" in s.kernel_name():
        assert out[0][4] == out[1][4] == out[2][4] == 120                  # split pipeline: one launch group per iteration, KLHR_CHUNK has no meaning
    else:
        assert out[0][4] == 1 and out[1][4] == 18 and out[2][4] == 3      # launches: 120 in one / ceil(120/7) / ceil(120/50)
    for o in out[1:]:
        for a, b in zip(o[:4], out[0][:4]):
            assert np.array_equal(a, b)


@pytest.mark.parametrize("name,variant", [("funnel10", 0), ("earnings", 0), ("funnel10", 2), ("funnel10", 1)])
def test_chain_results_do_not_depend_on_wavefront_company(oracle, name, variant):
    """A chain is a pure function of its seed and the configuration: its draws are bit-identical whether it runs alone,
    among 7 or among 63 other chains.  (The lane kernel lets idle lanes evaluate quadrature-node groups for chains that
    are still fitting — F = 2, 4 or 8 lanes per request depending on how many chains of the wavefront are left — and
    combines the partials in a fixed tree, so who evaluated what must not show in the result.)"""
    stem, data = model_spec(name)
    def run(seeds, n):
        s = make_sampler(stem, data, len(seeds), warmup=100, variant=variant)
        s.seed(np.asarray(seeds, dtype=np.uint64))
        s.step(n)
        return s.get("theta"), s.get("rng"), s.get("grad_evals")
    ref = run(np.arange(64), 130)
    for sub in (np.arange(8), np.arange(3, 30), np.array([5]), np.arange(40)):
        got = run(sub, 130)
        for a, b in zip(got, ref):
            assert np.array_equal(a, b[sub])


@pytest.mark.parametrize("name", ["normal2", "normal100"])
def test_mh_sampler_vs_golden(name):
    g = golden("g9_mh.npz")
    stem, data = model_spec(name)
    s = make_sampler(stem, data, 2)
    s.mh_seed(np.array([0, 1], dtype=np.uint64))
    th0 = s.get("theta")
    for k in (0, 1):
        assert np.array_equal(th0[k], g[f"{name}_s{k}_theta0"])
    s.mh_step(0.09, 2000, record_draws=True)
    dr = s.draws(0, 2000)
    for k in (0, 1):
        ref = g[f"{name}_s{k}_thetas"]
        mine = dr[::10, k] if name == "normal100" else dr[:, k]
        assert np.allclose(mine, ref, rtol=0, atol=1e-12)
        assert np.array_equal(s.get("rng")[k], g[f"{name}_s{k}_state1"])
        assert abs(s.get("acc")[k] - g[f"{name}_s{k}_acc"][0]) < 1e-12


def test_overrelaxed_proposal_vs_oracle(oracle):
    """a13: the over-relaxed MH step.  GPU and oracle share the RNG convention (K uniforms, then a + b uniforms), so the
    stream position is exact and z' agrees to the accuracy of Phi / Phi^-1 (ocml vs the oracle's erfc-based versions)."""
    g = golden("g8_funnel10_s0_w100.npz")
    M = 256
    stem, data = model_spec("funnel10")
    s = make_sampler(stem, data, M, warmup=100, overrelaxed=True)
    s.set("theta", g["theta_before"][:M])
    s.set("rng", g["state_before"][:M])
    cnt = np.zeros((M, 4), dtype=np.int64)
    cnt[:, 0] = np.arange(M)
    s.set("counters", cnt)
    s.metropolis(g["eta"][:M], g["rho"][:M])
    d = s.get("diag")
    from klhr_amd._lib import DIAG_COLS as DC
    col = {k: i for i, k in enumerate(DC)}
    th = s.get("theta")
    rng = s.get("rng")
    n_atom = 0
    for m in range(M):
        o = oracle.Chain(stem, data, seed=0, warmup=100, overrelaxed=True)
        o.set("theta", g["theta_before"][m]); o.set("rng", g["state_before"][m]); o.set("draw", m)
        o.metropolis(g["eta"][m], g["rho"][m])
        od = o.get("diag")
        assert np.array_equal(o.get("rng"), rng[m]), m
        assert abs(d[m, col["zp"]] - od["zp"]) <= 1e-9 * (1 + abs(od["zp"])), (m, d[m, col["zp"]], od["zp"])
        assert d[m, col["logu"]] == pytest.approx(od["logu"], rel=1e-15)
        if abs(od["logu"] - min(0.0, od["r"])) > 1e-6:
            assert d[m, col["accept"]] == od["accept"]
            assert np.allclose(th[m], o.get("theta"), rtol=1e-9, atol=1e-10)
        n_atom += abs(od["zp"]) < 1e-9
    assert 0 < n_atom < M          # both the u' = u branch and the beta branches were exercised


G11 = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "g11_klhr_*_w*.npz")))


@pytest.mark.parametrize("fn", G11)
def test_klhr_gaussian_variant_vs_oracle(oracle, fn):
    """the Gaussian-proposal sibling KLHR (klhr.py).  Every record of a reference trace becomes one GPU
    chain; direction, fit, MH step and closure-or-update (incl. the Smoother-driven K) are compared with the oracle."""
    parts = fn[len("g11_klhr_"):-4].split("_")
    name = parts[0]
    flags = {"eigen_method_one": False} if parts[-1] == "m2" else {}
    g = golden(fn)
    stem, data = model_spec(name)
    M, warmup, seed = min(int(g["M"][0]), 300), int(g["warmup"][0]), int(g["seed"][0])
    # ---- oracle pass (teacher-forced on the reference's thetas), snapshotting the state before every draw
    o = oracle.KLHRChain(stem, data, seed=seed, warmup=warmup, **flags)
    J = o.J
    fields = ["mean", "cov", "eigvecs", "eigvals", "om_m", "om_v", "pca_v"]
    before = {f: [] for f in fields}
    after = {f: [] for f in fields}
    cnt_b, cnt_a, ks_b, ks_a = [], [], [], []
    R = dict(rho=[], eta=[], diag=[], rng_dir=[], rng_fit=[], theta_mh=[], diag_mh=[], rng_mh=[])
    L = oracle.lib()
    for m in range(M):
        for f in fields:
            before[f].append(o.get(f).copy())
        cnt_b.append([m, o.get("om_N"), o.get("pca_n"), o.get("cidx")])
        o.set("theta", g["theta_before"][m]); o.set("rng", g["state_before"][m])
        R["rho"].append(o.random_direction()); R["rng_dir"].append(o.get("rng"))
        R["eta"].append(o.fit(g["rho"][m])); R["diag"].append(o.get("diag")); R["rng_fit"].append(o.get("rng"))
        o.set("draw", m)
        o.metropolis(np.concatenate([g["eta"][m], [0, 0]]), g["rho"][m])
        R["theta_mh"].append(o.get("theta")); R["diag_mh"].append(o.get("diag")); R["rng_mh"].append(o.get("rng"))
        o.set("theta", g["theta_after"][m])
        o.adapt()
        for f in fields:
            after[f].append(o.get(f).copy())
        cnt_a.append([m + 1, o.get("om_N"), o.get("pca_n"), o.get("cidx")])
        ks_a.append(L.klhr_oracle_get_K(o._h))
    # ---- device
    kw = dict(tol=1e-12, scale_clip=600, overrelaxed=False, eigen_method_one=flags.get("eigen_method_one", True), J=J, variant=1)
    s = make_sampler(stem, data, M, warmup=warmup, **kw)
    s.set("theta", g["theta_before"][:M]); s.set("rng", g["state_before"][:M])
    for f in fields:
        s.set(f, np.array(before[f]))
    s.set("counters", np.array(cnt_b, dtype=np.int64))
    rho = s.random_direction()
    assert np.array_equal(s.get("rng"), np.array(R["rng_dir"]))
    assert np.array_equal(rho, np.array(R["rho"])), np.abs(rho - np.array(R["rho"])).max()
    eta = s.fit(g["rho"][:M])
    assert np.array_equal(s.get("rng"), np.array(R["rng_fit"]))          # exactly one normal consumed (klhr.py:128)
    d = s.get("diag")
    from klhr_amd._lib import DIAG_COLS as DC
    col = {k: i for i, k in enumerate(DC)}
    o_nfev1 = np.array([x["nfev1"] for x in R["diag"]]); o_nfev2 = np.array([x["nfev2"] for x in R["diag"]])
    o_st2 = np.array([x["status2"] for x in R["diag"]])
    same = (d[:, col["nfev1"]] == o_nfev1) & (d[:, col["nfev2"]] == o_nfev2)
    conv = o_st2 == 0
    assert (same & conv).sum() >= 0.98 * conv.sum(), ((same & conv).sum(), conv.sum())
    oe = np.array(R["eta"])
    err = np.abs(eta[:, :2] - oe[:, :2]) / (1 + np.abs(oe[:, :2]))
    assert np.all(err[same & conv] < 2e-6), err[same & conv].max()
    assert np.all(eta[:, 2:] == 0)
    # MH step from the reference's eta
    cb = np.array(cnt_b, dtype=np.int64)
    s.set("counters", cb)
    e4 = np.zeros((M, 4)); e4[:, :2] = g["eta"][:M]
    s.metropolis(e4, g["rho"][:M])
    d = s.get("diag")
    assert np.array_equal(s.get("rng"), np.array(R["rng_mh"]))
    o_zp = np.array([x["zp"] for x in R["diag_mh"]]); o_r = np.array([x["r"] for x in R["diag_mh"]])
    o_logu = np.array([x["logu"] for x in R["diag_mh"]]); o_acc = np.array([x["accept"] for x in R["diag_mh"]])
    assert np.all(np.abs(d[:, col["zp"]] - o_zp) <= 1e-12 * (1 + np.abs(o_zp)))
    assert np.all(np.abs(d[:, col["r"]] - o_r) <= 1e-9 * (1 + np.abs(o_r)))
    clear = np.abs(o_logu - np.minimum(0.0, o_r)) > 1e-8
    assert np.array_equal(d[clear, col["accept"]], o_acc[clear])
    # closure-or-update from the reference's theta: bit-exact, and K follows the Smoother (klhr.py:212-221)
    moved = np.any(s.get("theta") != g["theta_before"][:M], axis=1)
    s.set("theta", g["theta_after"][:M])
    cb2 = cb.copy(); cb2[:, 0] += 1
    s.set("counters", cb2)
    # Smoother state before each record comes from the oracle's run: rebuild it from the accept history
    ks = np.zeros((M, 4)); x, cntr, K = 10.0, 0.0, 10
    closures = set([50, 100])
    for m in range(M):
        ks[m] = [x, cntr, K, float(bool(g["accept"][m]) and np.any(g["theta_after"][m] != g["theta_before"][m]))]
        if (m + 1) in closures:
            K = int(min(max(x, 1.0), 50.0)); x, cntr = 10.0, 0.0
        else:
            cntr += 1.0; kk = cntr ** -0.75
            x = kk * (x + (1.0 if ks[m][3] else -1.0)) + (1 - kk) * x
    s.set("kstate", ks)
    s.adapt()
    assert np.array_equal(s.get("counters"), np.array(cnt_a, dtype=np.int64))
    for f in fields:
        assert np.array_equal(s.get(f), np.array(after[f])), f
    assert np.array_equal(s.get("kstate")[:, 2].astype(int), np.array(ks_a)), "K after adapt"
    assert np.array_equal(np.array(ks_a), g["K"][:M])


def test_klhr_free_running_vs_reference():
    g = golden("g11_klhr_funnel10_s0_w100.npz")
    stem, data = model_spec("funnel10")
    s = make_sampler(stem, data, 1, warmup=100, tol=1e-12, scale_clip=600, eigen_method_one=True, J=2, variant=1)
    s.seed(np.array([0], dtype=np.uint64))
    s.step(120, record_draws=True)
    dr = s.draws(0, 120)[:, 0]
    err = np.abs(dr - g["theta_after"][:120]).max(axis=1)
    assert err[:50].max() < 1e-9
    assert int(s.get("kstate")[0, 2]) == g["K"][119]


G12 = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "g12_sub_*_w*.npz")))


@pytest.mark.parametrize("fn", G12)
def test_sub_variant_vs_oracle(oracle, fn):
    """SUBKLHRSINH (3-parameter proposal).  Per-record teacher forcing of fit and MH step against the oracle."""
    name = fn[len("g12_sub_"):-4].split("_")[0]
    g = golden(fn)
    stem, data = model_spec(name)
    M, warmup = min(int(g["M"][0]), 300), int(g["warmup"][0])
    # records before the first window closure share the initial adaptation state (mean 0, cov 1, no eigenvectors)
    M = min(M, 49)
    s = make_sampler(stem, data, M, warmup=warmup, variant=2)
    s.set("theta", g["theta_before"][:M]); s.set("rng", g["state_before"][:M])
    cnt = np.zeros((M, 4), dtype=np.int64); cnt[:, 0] = np.arange(M)
    s.set("counters", cnt)
    rho = s.random_direction()
    assert np.allclose(rho, g["rho"][:M], rtol=1e-13, atol=1e-15)    # pre-closure direction: pure RNG + normalisation
    eta = s.fit(g["rho"][:M])
    assert np.array_equal(s.get("rng"), g["state_mh"][:M])   # 1 + 3 normals consumed
    d = s.get("diag")
    from klhr_amd._lib import DIAG_COLS as DC
    col = {k: i for i, k in enumerate(DC)}
    o_eta, o_n1, o_n2 = [], [], []
    for m in range(M):
        o = oracle.Chain(stem, data, seed=0, warmup=warmup, variant="sub")
        o.set("theta", g["theta_before"][m]); o.set("rng", g["state_before"][m])
        o.random_direction()
        o_eta.append(o.fit(g["rho"][m])); dd = o.get("diag"); o_n1.append(dd["nfev1"]); o_n2.append(dd["nfev2"])
    o_eta = np.array(o_eta)
    same = (d[:, col["nfev1"]] == np.array(o_n1)) & (d[:, col["nfev2"]] == np.array(o_n2))
    assert same.mean() >= 0.95, same.mean()
    err = np.abs(eta[:, :3] - o_eta[:, :3]) / (1 + np.abs(o_eta[:, :3]))
    assert np.all(err[same] < 2e-6) and np.all(eta[:, 3] == 0)
    assert np.array_equal(d[:, col["init2"]], g["init"][:M, 2]) and np.all(d[:, col["init3"]] == 0)
    s.set("counters", cnt)
    e4 = np.zeros((M, 4)); e4[:, :3] = g["eta"][:M]
    s.metropolis(e4, g["rho"][:M])
    d = s.get("diag")
    assert np.array_equal(s.get("rng"), g["state_after"][:M])
    assert np.all(np.abs(d[:, col["zp"]] - g["zp"][:M]) <= 1e-11 * (1 + np.abs(g["zp"][:M])))
    assert np.all(np.abs(d[:, col["r"]] - g["r"][:M]) <= 1e-9 * (1 + np.abs(g["r"][:M])))
    clear = np.abs(g["logu"][:M] - np.minimum(0.0, g["r"][:M])) > 1e-8
    assert np.array_equal(d[clear, col["accept"]].astype(bool), g["accept"][:M][clear])


def test_sub_free_running_vs_reference():
    g = golden("g12_sub_funnel10_s0_w100.npz")
    stem, data = model_spec("funnel10")
    s = make_sampler(stem, data, 1, warmup=100, variant=2)
    s.seed(np.array([0], dtype=np.uint64))
    s.step(120, record_draws=True)
    dr = s.draws(0, 120)[:, 0]
    err = np.abs(dr - g["theta_after"][:120]).max(axis=1)
    assert err[:50].max() < 1e-9
    bad = np.flatnonzero(err > 1e-6)
    first = int(bad[0]) if bad.size else 120
    assert np.array_equal(np.any(dr[1:first] != dr[:first - 1], axis=1), g["accept"][1:first])


G13 = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "g13_slice_*_w*.npz")))


def parse_g13(fn):
    parts = fn[len("g13_slice_"):-4].split("_")
    flags = {}
    if "m2" in parts[3:]:
        flags["eigen_method_one"] = False
    if "s" in parts[3:]:
        flags["scale_dir_cov"] = True
    return parts[0], flags


@pytest.mark.parametrize("fn", G13)
def test_slice_variant_vs_oracle_and_reference(oracle, fn):
    """Slice sampler (slice.py): every record of a golden trace becomes one GPU chain.  Direction draw and
    streaming update are bit-exact vs the oracle as for the other variants; the slice step from the reference's rho must
    consume the generator exactly like the reference (same number of stepping-out / shrinkage draws) and land on the same
    point, except where a comparison g(x) <= log y is decided by the last bits of the log density."""
    name, flags = parse_g13(fn)
    g = golden(fn)
    stem, data = model_spec(name)
    M, warmup, seed = int(g["M"][0]), int(g["warmup"][0]), int(g["seed"][0])
    M = min(M, 300)
    # oracle, teacher-forced, to have the adaptation state before every record
    c = oracle.SliceChain(stem, data, seed=seed, warmup=warmup, **flags)
    fields = ["mean", "cov", "eigvecs", "eigvals", "om_m", "om_v", "pca_v"] + (["omd_m", "omd_v"] if flags.get("scale_dir_cov") else [])
    before = {f: [] for f in fields}
    after = {f: [] for f in fields}
    cnt_before, cnt_after, rho_o, rng_dir, th_o, rng_o, ev_o = [], [], [], [], [], [], []
    for m in range(M):
        for f in fields:
            before[f].append(c.get(f).copy())
        cnt_before.append([m, c.get("om_N"), c.get("pca_n"), c.get("cidx")])
        c.set("theta", g["theta_before"][m]); c.set("rng", g["state_before"][m])
        rho_o.append(c.random_direction()); rng_dir.append(c.get("rng"))
        e0 = c.slice_evals()
        c.set("draw", m)
        th_o.append(c.uni_slice(g["rho"][m])); rng_o.append(c.get("rng")); ev_o.append(c.slice_evals() - e0)
        c.set("theta", g["theta_after"][m]); c.set("rng", g["state_after"][m])
        c.adapt()
        for f in fields:
            after[f].append(c.get(f).copy())
        cnt_after.append([m + 1, c.get("om_N"), c.get("pca_n"), c.get("cidx")])
    s = make_sampler(stem, data, M, warmup=warmup, tol=1e-12, eigen_method_one=flags.get("eigen_method_one", True),
                     scale_dir_cov=flags.get("scale_dir_cov", False), variant=3)
    assert s.kernel_name() == "klhr_iter_kernel"
    s.set("theta", g["theta_before"][:M]); s.set("rng", g["state_before"][:M])
    for f in fields:
        s.set(f, np.array(before[f]))
    s.set("counters", np.array(cnt_before, dtype=np.int64))
    rho = s.random_direction()
    assert np.array_equal(s.get("rng"), np.array(rng_dir))
    assert np.array_equal(rho, np.array(rho_o)), np.abs(rho - np.array(rho_o)).max()
    # ---- the slice step from the reference's rho
    ge0 = s.get("grad_evals").copy()
    s.slice_step(g["rho"][:M])
    th, rng, ev = s.get("theta"), s.get("rng"), s.get("grad_evals") - ge0
    same_o = np.all(rng == np.array(rng_o), axis=1) & (ev == np.array(ev_o))
    same_r = np.all(rng == g["state_after"][:M], axis=1) & (ev == g["evals"][:M])
    assert same_o.mean() >= 0.99 and same_r.mean() >= 0.99, (same_o.mean(), same_r.mean())
    assert np.allclose(th[same_o], np.array(th_o)[same_o], rtol=1e-12, atol=1e-13)
    assert np.allclose(th[same_r], g["theta_after"][:M][same_r], rtol=1e-12, atol=1e-13)
    assert np.all(s.get("counters")[:, 0] == np.arange(1, M + 1))
    assert not s.get("status").any()
    # ---- closure-or-update from the reference's theta: bit-exact vs the oracle
    s.set("theta", g["theta_after"][:M]); s.set("rng", g["state_after"][:M])
    s.adapt()
    for f in fields:
        if f.startswith("omd"):      # moments of the model gradient: device libm (exp) differs from glibc by ulps
            assert np.allclose(s.get(f), np.array(after[f]), rtol=1e-12, atol=1e-13), f
        else:
            assert np.array_equal(s.get(f), np.array(after[f])), f
    assert np.array_equal(s.get("counters"), np.array(cnt_after, dtype=np.int64))


def test_slice_free_running_and_python_class():
    """klhr_amd.Slice with the reference's constructor: its draw() loop follows the reference's own draw() until the
    first window closure (nothing depends on CCIPCA before), and keeps sampling the right distribution after."""
    from klhr_amd import BSModel, Slice
    stan = os.path.join(os.path.dirname(GOLDEN), "..", "klhr_amd", "stan")
    g = golden("g13_slice_funnel10_s0_w100.npz")
    m = BSModel(os.path.join(stan, "funnel.stan"), os.path.join(stan, "funnel.json"))
    algo = Slice(m, seed=0, warmup=100)
    th = np.array([algo.draw() for _ in range(50)])
    assert np.allclose(th, g["theta_after"][:50], rtol=1e-11, atol=1e-12)
    assert algo.grad_evals == int(g["evals"][:50].sum())
    assert algo.acceptance_probability == pytest.approx(1.0)
    with pytest.raises(NotImplementedError):
        Slice(m, seed=0, m=5)
    # many chains on a 2-D standard normal: moments of the draws
    mn = BSModel(os.path.join(stan, "normal.stan"), os.path.join(stan, "normal.json"))
    many = Slice(mn, seed=3, warmup=100, chains=512)
    d = many.draws(400)[150:]
    assert abs(d.mean()) < 0.02 and abs(d.var() - 1.0) < 0.04


@pytest.mark.parametrize("name", ["funnel100", "illnormal100", "funnel10", "ar1x100"])
@pytest.mark.parametrize("variant,over", [(0, False), (0, True), (1, False), (2, False)])
def test_node_parallel_fit_equals_lane_per_chain_fit(monkeypatch, kernel_mode, name, variant, over):
    """Split pipeline: up to 32 768 chains both stages of fit() run in klhr_fitn_kernel (8 lanes per chain, one quadrature
    node per lane, butterfly sums with the lane form's tree), above that in klhr_fit12_kernel (lane = chain).
    Which one ran must not show in any bit of a chain's state — size independence (test_gpu_fullsize.py) rests on it.
    The small clip bounds make the exact element-wise patch of the closed form (rare at the defaults) a common path."""
    if kernel_mode != "split":
        pytest.skip("the fit kernels of the split pipeline")
    stem, data = model_spec(name)
    C = 203
    seeds = np.arange(C, dtype=np.uint64) + 4100
    for clips in ({}, {"grad_clip": 3.0, "scale_clip": 3.0}):
        out = []
        monkeypatch.setenv("KLHR_FUSED", "0")
        for mode in ("0", "1"):
            monkeypatch.setenv("KLHR_FITN", mode)
            s = make_sampler(stem, data, C, warmup=100, variant=variant, overrelaxed=over, **clips)
            s.seed(seeds)
            s.step(70, record_diag=True)
            assert ("fitn" in s.kernel_name()) == (mode == "1"), s.kernel_name()
            out.append([s.get(f) for f in ("theta", "rng", "cov", "eigvecs", "pca_v", "grad_evals", "acc", "counters", "diag", "status")])
        for a, b in zip(out[0], out[1]):
            assert np.array_equal(a, b, equal_nan=True)


@pytest.mark.parametrize("name", ["funnel100", "illnormal100", "ar1x100"])
@pytest.mark.parametrize("variant,over,flags", [(0, False, {}), (0, True, {}), (1, False, {}), (2, False, {}),
                                                (0, False, {"scale_dir_cov": True}), (0, False, {"eigen_method_one": True})])
def test_paired_streaming_wavefronts_equal_single_wavefront(monkeypatch, kernel_mode, name, variant, over, flags):
    """Split pipeline, 16-chain tiles: the streaming kernel runs a tile's tail and head on two wavefronts (RNG state handed
    over in LDS after the MH draws, three workgroup barriers; klhr_split.hpp) — KLHR_STREAM_PAIR=0 runs them one after
    the other on one wavefront.  120 draws cross both window closures of warmup = 100, where the head waits for the tail;
    chain counts that leave the last tile / workgroup partly empty."""
    if kernel_mode != "split":
        pytest.skip("the streaming kernel of the split pipeline")
    if name == "ar1x100" and flags.get("scale_dir_cov"):
        pytest.skip("ar1 with scale_dir_cov runs on the octet kernel")
    stem, data = model_spec(name)
    for C in (203, 16, 33):
        seeds = np.arange(C, dtype=np.uint64) + 5200
        out = []
        monkeypatch.setenv("KLHR_FUSED", "0")
        for mode in ("0", "1"):
            monkeypatch.setenv("KLHR_STREAM_PAIR", mode)
            s = make_sampler(stem, data, C, warmup=100, variant=variant, overrelaxed=over, **flags)
            s.seed(seeds)
            s.step(120, record_draws=True)
            out.append([s.draws(0, 120)] + [s.get(f) for f in ("theta", "rng", "mean", "cov", "eigvecs", "eigvals", "om_m", "om_v", "pca_v",
                                                                 "grad_evals", "acc", "counters", "status")])
        for a, b in zip(out[0], out[1]):
            assert np.array_equal(a, b, equal_nan=True)


@pytest.mark.parametrize("name", ["funnel100", "illnormal100", "funnel10"])
@pytest.mark.parametrize("variant,over,flags", [(0, False, {}), (0, True, {}), (1, False, {}), (2, True, {}),
                                                (0, False, {"scale_dir_cov": True}), (0, False, {"eigen_method_one": True})])
def test_whole_iterations_per_tile_equal_the_kernel_sequence(monkeypatch, kernel_mode, name, variant, over, flags):
    """Split pipeline, up to 16 384 chains: klhr_step runs all draws of a call in ONE launch, a workgroup looping over the
    iterations of its 16-chain tile (klhr_tile_kernel: the streaming body in pair mode, then the node-parallel fit body);
    KLHR_FUSED=0 runs the same bodies as a sequence of launches.  Two calls of 70 and 50 draws (the second one starts with a
    head on a state the first one left), recorded draws, chain counts that leave a tile partly empty."""
    if kernel_mode != "split":
        pytest.skip("the split pipeline")
    stem, data = model_spec(name)
    for C in (203, 16, 33):
        seeds = np.arange(C, dtype=np.uint64) + 6400
        out = []
        for mode in ("0", "1"):
            monkeypatch.setenv("KLHR_FUSED", mode)
            s = make_sampler(stem, data, C, warmup=100, variant=variant, overrelaxed=over, **flags)
            assert ("tile" in s.kernel_name()) == (mode == "1"), s.kernel_name()
            s.seed(seeds)
            s.step(70, record_draws=True)
            d1 = s.draws(0, 70)
            s.step(50, record_draws=True)
            d2 = s.draws(0, 50)
            if C == 16 and name != "funnel100":
                s.step(300, record_draws=True)                  # more than one launch of the tile kernel (256 draws each)
                d2 = np.concatenate([d2, s.draws(0, 300)])
            out.append([d1, d2] + [s.get(f) for f in ("theta", "rng", "mean", "cov", "eigvecs", "eigvals", "om_m", "om_v", "pca_v",
                                                       "grad_evals", "acc", "counters", "status")])
        for a, b in zip(out[0], out[1]):
            assert np.array_equal(a, b, equal_nan=True)


@pytest.mark.parametrize("name", ["funnel100", "illnormal100", "ar1x100"])
def test_fit_kernel_forms_agree_over_a_population(monkeypatch, kernel_mode, name):
    """2 048 chains x 30 draws through the lane = chain fit kernels, the node-parallel fit kernel and the whole-iterations-per-tile
    kernel: the same bits.  (With the optimiser compiled under -ffp-contract=fast the back end fused a multiplication and an
    addition in one inlined copy and not in the other: one chain-draw in ~2 000 got a request point one ulp apart — too rare
    for the 203-chain tests above, so this one looks at a population.)"""
    if kernel_mode != "split":
        pytest.skip("the fit kernels of the split pipeline")
    stem, data = model_spec(name)
    C = 2048
    seeds = np.arange(C, dtype=np.uint64) + 1000
    out = []
    for fitn, fused in (("0", "0"), ("1", "0"), ("1", "1")):
        monkeypatch.setenv("KLHR_FITN", fitn)
        monkeypatch.setenv("KLHR_FUSED", fused)
        s = make_sampler(stem, data, C, warmup=100)
        s.seed(seeds)
        s.step(30)
        out.append([s.get(f) for f in ("theta", "rng", "eta", "pca_v", "grad_evals", "acc")])
    for o in out[1:]:
        for a, b in zip(out[0], o):
            assert np.array_equal(a, b, equal_nan=True)


SPLIT_FORMS = {"fit12": {"KLHR_KERNEL": "split", "KLHR_FITN": "0", "KLHR_FUSED": "0"}, "fitn": {"KLHR_KERNEL": "split", "KLHR_FITN": "1", "KLHR_FUSED": "0"},
               "tile": {"KLHR_KERNEL": "split", "KLHR_FITN": "1", "KLHR_FUSED": "1"}, "mono": {"KLHR_KERNEL": "mono"}}


@pytest.mark.parametrize("name", ["funnel10", "funnel100", "illnormal100", "ar1x100"])
@pytest.mark.parametrize("form", ["fit12", "fitn", "tile", "mono"])
def test_exact_patch_draws_vs_oracle(oracle, monkeypatch, kernel_mode, name, form):
    """The closed-form kernels' exact element-wise patch INSIDE THE FIT, against the oracle.  With scale_clip = grad_clip = 3 the
    clip of _logp_grad (klhr_sinh.py:158-161) bites at most quadrature nodes, so line_point's bound fails and nearly every KL
    evaluation of a fit takes the patch (at the defaults: 1 in 2 000-6 000).  (a) Free-running: 24 chains x 49 draws against
    oracle.Chain(scale_clip=3, grad_clip=3) — funnel / ill-normal follow the oracle to 1e-9 over all 49 draws (a chain may leave
    at a knife-edge decision: at most 2 of 24, and the element-wise octet kernel loses the same ones); on ar1 the KL of a
    Gaussian target is flat in the tail parameters and free-running chains part from ANY other arithmetic within ~5 draws
    (the octet kernel, which has no closed form, does too), so there (b) carries the weight.  (b) Teacher-forced: every draw of two
    oracle trajectories (2 x 60 records, both window closures) is handed to its own device chain with the oracle's state before
    it; one whole draw through the production kernels; generator position identical, theta' within 1e-9 in >= 95 % of the records
    and never off by more than a move along the fitted proposal (accept decisions on a knife edge)."""
    if kernel_mode != "auto":
        pytest.skip("this test selects every form itself")
    for k, v in SPLIT_FORMS[form].items():
        monkeypatch.setenv(k, v)
    stem, data = model_spec(name)
    kw = dict(warmup=100, scale_clip=3.0, grad_clip=3.0)
    # ---- (a) free-running
    C, n = 24, 49
    seeds = np.arange(C, dtype=np.uint64) + 3100
    ref = np.array([oracle.Chain(stem, data, seed=int(sd), **kw).draw_many(n) for sd in seeds]).transpose(1, 0, 2)
    s = make_sampler(stem, data, C, **kw)
    if form != "mono":
        assert ("tile" in s.kernel_name()) == (form == "tile") and ("fitn" in s.kernel_name()) == (form == "fitn"), s.kernel_name()
    s.seed(seeds)
    s.step(n, record_draws=True)
    got = s.draws(0, n)
    err = np.abs(got - ref).max(axis=2)
    first = np.array([np.argmax(err[:, c] > 1e-9) if (err[:, c] > 1e-9).any() else n for c in range(C)])
    if stem != "ar1":
        assert (first == n).sum() >= C - 2, (name, form, sorted(first))
    else:
        assert np.median(first) >= 4, (name, form, sorted(first))
    assert not s.get("status").any()
    # ---- (b) teacher-forced whole draws
    recs = [oracle_state_trace(oracle, stem, data, seed=sd, n=60, **kw) for sd in (4200, 4201)]
    M = sum(r["theta"].shape[0] for r in recs)
    cat = lambda f: np.concatenate([r[f] for r in recs])
    s = make_sampler(stem, data, M, **kw)
    for f in STATE_FIELDS:
        s.set(f, cat(f))
    s.set("counters", cat("counters"))
    s.set("acc", cat("acc_before"))
    s.step(1)
    assert np.array_equal(s.get("rng"), cat("rng_after"))
    th, want = s.get("theta"), cat("theta_after")
    e = np.abs(th - want).max(axis=1) / (1.0 + np.abs(want).max(axis=1))
    assert np.mean(e < 1e-9) >= 0.95, (name, form, np.sort(e)[-8:])
    moved = np.abs(want - cat("theta")).max(axis=1) + np.abs(th - cat("theta")).max(axis=1)
    assert np.all(e <= 1e-9 + 2.0 * moved), (name, form)


@pytest.mark.parametrize("form", ["tile", "fitn"])
@pytest.mark.parametrize("stem,D", [("funnel", 2), ("normal", 2), ("funnel", 4), ("funnel", 26), ("ill-normal", 30), ("ar1", 33), ("normal", 41), ("ill-normal", 103)])
def test_quad_mode_ragged_dimensions(oracle, monkeypatch, kernel_mode, stem, D, form):
    """Quad mode of the 16-chain tiles (four lanes per chain, coordinate i = 4 step + q; klhr_split.hpp) at dimensions that are NOT
    multiples of four (a last, partial step; no whole step at dimension 2 and 3; the funnel's dimension is its data's D + 1: 3, 5, 27), on every separable target, through two window closures (the
    rank rule, the column norms and the direction cache are written by four lanes per chain) and with the direction's normals
    drawn four per round from the generator's stride-4 recurrence (device_rng.hpp: rng_std_normal_quad).
    (a) Against lane = chain (64-chain tiles, the same pipeline without quad mode): every draw, the generator and the whole
    adaptation state are the SAME BITS.  (b) Against the oracle up to the first closure: generator position exact, theta within 1e-9
    (beyond a closure the comparison says nothing about the mapping: with 20-draw windows CCIPCA's first updates — weights
    (n - 1 - l) / n < 0 for n <= 4 — amplify the fit's rounding-level differences to 1e-6 in the eigenvectors, for the monolithic
    kernels just the same)."""
    if kernel_mode != "auto":
        pytest.skip("this test selects the forms itself")
    data = {"N": D} if stem == "ar1" else {"D": D}
    kw = dict(warmup=60, windowsize=20)
    C, n = 20, 70
    seeds = np.arange(C, dtype=np.uint64) + 5200 + D
    fields = STATE_FIELDS + ["acc", "grad_evals", "counters"]

    def run(env, nd):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        s = make_sampler(stem, data, C, **kw)
        s.seed(seeds)
        s.step(nd, record_draws=True)
        assert not s.get("status").any()
        return s.kernel_name(), s.draws(0, nd), {f: s.get(f) for f in fields}

    name64, d64, st64 = run({"KLHR_KERNEL": "split", "KLHR_FITN": "1", "KLHR_FUSED": "0", "KLHR_TILE": "64"}, n)
    assert "klhr_stream_kernel" in name64, name64
    nameq, dq, stq = run(dict(SPLIT_FORMS[form], KLHR_TILE="16"), n)
    assert ("tile" in nameq) == (form == "tile") and ("stream16" in nameq) == (form != "tile"), nameq
    assert np.array_equal(dq, d64), (stem, D, form, int(np.argmax((dq != d64).any(axis=(1, 2)))))
    for f in fields:
        assert np.array_equal(stq[f], st64[f]), (stem, D, form, f)
    # (b) the oracle, up to the first closure (draw 20)
    m = 19
    chains = [oracle.Chain(stem, data, seed=int(sd), **kw) for sd in seeds]
    ref = np.array([c.draw_many(m) for c in chains]).transpose(1, 0, 2)
    _, dm, stm = run(dict(SPLIT_FORMS[form], KLHR_TILE="16"), m)
    assert np.array_equal(stm["rng"], np.array([c.get("rng") for c in chains])), (stem, D, form)
    err = np.abs(dm - ref).max(axis=2) / (1.0 + np.abs(ref).max(axis=2))
    bad = (err > 1e-9).any(axis=0)
    assert bad.sum() <= (2 if stem != "ar1" else C // 2), (stem, D, form, err.max(axis=0))


@pytest.mark.parametrize("opts", [dict(variant=1), dict(variant=1, overrelaxed=True), dict(variant=2), dict(overrelaxed=True), dict(scale_dir_cov=True),
                                  dict(eigen_method_one=True), dict(J=3), dict(J=1)],
                         ids=["klhr", "klhr-overrelaxed", "sub", "overrelaxed", "scale_dir_cov", "eigen_method_one", "J3", "J1"])
@pytest.mark.parametrize("stem,D", [("funnel", 26), ("ill-normal", 30)])
def test_quad_mode_variants_and_flags_equal_lane_per_chain(monkeypatch, kernel_mode, stem, D, opts):
    """Every sampler variant and flag of the split pipeline through quad mode (16-chain tiles: the tile kernel AND the kernel sequence
    in pair mode) at a dimension with a partial last step, against lane = chain (64-chain tiles): 70 draws through two closures, the
    same draws, generator and adaptation state bit for bit.  (KLHR's Smoother state, the gradient's OnlineMoments of scale_dir_cov,
    rng.choice()'s uniform ahead of the quad generator's rounds, one and three CCIPCA components.)"""
    if kernel_mode != "auto":
        pytest.skip("this test selects the forms itself")
    data = {"D": D}
    kw = dict(warmup=60, windowsize=20, **opts)
    C, n = 20, 70
    seeds = np.arange(C, dtype=np.uint64) + 6100 + D
    fields = STATE_FIELDS + ["acc", "grad_evals", "counters"]

    def run(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        s = make_sampler(stem, data, C, **kw)
        s.seed(seeds)
        s.step(n, record_draws=True)
        return s.kernel_name(), s.draws(0, n), {f: s.get(f) for f in fields}, s.get("status")

    name64, d64, st64, status64 = run({"KLHR_KERNEL": "split", "KLHR_FITN": "1", "KLHR_FUSED": "0", "KLHR_TILE": "64"})
    assert "klhr_stream_kernel" in name64, name64
    for form in ("tile", "fitn"):
        nameq, dq, stq, statusq = run(dict(SPLIT_FORMS[form], KLHR_TILE="16"))
        assert ("tile" in nameq) == (form == "tile") and ("stream16" in nameq) == (form != "tile"), nameq
        assert np.array_equal(statusq, status64), (stem, D, opts, form)
        assert np.array_equal(dq, d64, equal_nan=True), (stem, D, opts, form, int(np.argmax((dq != d64).any(axis=(1, 2)))))
        for f in fields:
            assert np.array_equal(stq[f], st64[f], equal_nan=True), (stem, D, opts, form, f)


@pytest.mark.parametrize("opts", [dict(), dict(variant=1), dict(variant=1, overrelaxed=True), dict(variant=2), dict(overrelaxed=True), dict(eigen_method_one=True),
                                  dict(scale_clip=3.0, grad_clip=3.0)],
                         ids=["sinh", "klhr", "klhr-overrelaxed", "sub", "overrelaxed", "eigen_method_one", "clip3"])
@pytest.mark.parametrize("stem,D", [("funnel", 9), ("funnel", 1), ("ill-normal", 26), ("normal", 3)])
def test_lane_kernel_equals_split_pipeline(monkeypatch, kernel_mode, stem, D, opts):
    """Dimension <= 26, separable targets: the lane-per-chain kernel and the split pipeline (tile kernel, kernel sequence on 16-chain and on
    64-chain tiles with the lane = chain fit) return the SAME BITS — draws, generator, adaptation state, evaluation counts — over 160 draws
    through both closures of warmup = 100.  This is what lets klhr_create choose the mapping by the chain count (tile kernel up to 16 384
    chains, lane kernel above) without the choice showing in a chain.  Two things make it so: the lane kernel takes its line sums from the
    unnormalised direction and scales them by 1 / ||r||, as the streaming kernel does, and every form of the KL evaluation replaces the
    closed form by the element-wise value for exactly the nodes whose own clip bound fails (clip3: nearly every evaluation)."""
    if kernel_mode != "auto":
        pytest.skip("this test selects the mappings itself")
    data = {"D": D}
    kw = dict(warmup=100, **opts)
    C, n = 72, 160
    seeds = np.arange(C, dtype=np.uint64) + 7300 + D
    fields = STATE_FIELDS + ["acc", "grad_evals", "counters", "status"]

    def run(env):
        for k in ("KLHR_KERNEL", "KLHR_FITN", "KLHR_FUSED", "KLHR_TILE"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        s = make_sampler(stem, data, C, **kw)
        s.seed(seeds)
        s.step(n, record_draws=True)
        return s.kernel_name(), s.draws(0, n), {f: s.get(f) for f in fields}

    name0, d0, st0 = run({"KLHR_KERNEL": "lane"})
    assert name0 == "klhr_lane_kernel", name0
    for env in ({}, {"KLHR_KERNEL": "split", "KLHR_FUSED": "0"}, {"KLHR_KERNEL": "split", "KLHR_FUSED": "0", "KLHR_TILE": "64", "KLHR_FITN": "0"}):
        name, d, st = run(env)
        assert "lane" not in name, (name, env)          # (the default mapping at 72 chains is the tile kernel)
        assert np.array_equal(d, d0, equal_nan=True), (stem, D, opts, name, int(np.argmax((d != d0).any(axis=(1, 2)))))
        for f in fields:
            assert np.array_equal(st[f], st0[f], equal_nan=True), (stem, D, opts, name, f)

"""GPU tests of the Python host layer that mirrors the reference's classes (klhr_amd/{bsmodel,klhr_sinh,mh,mcmc}.py):
a user of the reference's API gets the reference's numbers."""
import os

import numpy as np
import pytest

from conftest import ROOT, golden

pytestmark = pytest.mark.gpu
STAN = os.path.join(ROOT, "klhr_amd", "stan")


def funnel_model():
    from klhr_amd import BSModel
    return BSModel(os.path.join(STAN, "funnel.stan"), os.path.join(STAN, "funnel.json"))


def test_bsmodel_surface_on_device():
    g = golden("g4_kl.npz")
    m = funnel_model()
    th = g["funnel10_theta"]
    for i in (0, 3, 9):
        lp, grad = m.log_density_gradient(th[i])
        assert isinstance(lp, float) and grad.shape == (10,)
        assert abs(lp - g["funnel10_lp"][i]) <= 1e-12 * abs(g["funnel10_lp"][i])
        assert np.allclose(grad, g["funnel10_lpgrad"][i], rtol=1e-12)
        assert m.log_density(th[i]) == lp
    lp_all = m.log_density(th)
    assert lp_all.shape == (th.shape[0],)
    lp, H = m.log_density_hessian(th[0])
    assert H.shape == (10, 10) and np.allclose(H, H.T)
    ev = np.exp(-th[0][0])
    assert np.allclose(np.diag(H)[1:], -ev, rtol=1e-5)           # d2/dalpha2 = -exp(-v)
    assert m.Hamiltonian(th[0], np.ones(10)) == pytest.approx(-lp + 5.0)
    # error convention of bsmodel.py:15-30: failure -> -inf and zero gradient
    bad = th[0].copy()
    bad[0] = np.nan
    lp, grad = m.log_density_gradient(bad)
    assert lp == -np.inf and not grad.any()


def test_klhrsinh_single_chain_is_the_reference_chain():
    from klhr_amd import KLHRSINH
    g = golden("g8_public_api.npz")
    g0 = golden("g8_funnel10_s0_w100.npz")
    algo = KLHRSINH(funnel_model(), seed=0, warmup=100, overrelaxed=False)
    assert algo.D == 10 and algo.theta.shape == (10,)
    assert np.array_equal(algo.theta, g0["theta0"])
    st = algo.rng.bit_generator.state["state"]
    assert (st["state"] >> 64, st["state"] & (2**64 - 1)) == (int(g0["state_before"][0][0]), int(g0["state_before"][0][1]))
    th = np.array([algo.draw() for _ in range(40)])
    assert th.shape == (40, 10)
    assert np.abs(th - g["s0_thetas"][:40]).max() < 1e-9
    assert algo.acceptance_probability.shape == (1,)
    assert algo.acceptance_probability[0] == pytest.approx(g0["accprob"][39], abs=1e-12)
    assert algo.grad_evals == g0["grad_evals"][39]
    more = np.array([next(algo) for _ in range(3)])          # iterator protocol (mcmc.py:19-23)
    assert np.abs(more - g["s0_thetas"][40:43]).max() < 1e-9


def test_sample_row0_is_initial_point():
    from klhr_amd import BSModel, KLHRSINH
    g = golden("g8_public_api.npz")
    m = BSModel(os.path.join(STAN, "normal.stan"), os.path.join(STAN, "normal.json"))
    algo = KLHRSINH(m, seed=3, warmup=100, overrelaxed=False)
    init = algo.theta.copy()
    s = algo.sample(50)
    assert s.shape == (50, 2) and np.array_equal(s[0], init)
    assert np.abs(s - g["sample_normal2"]).max() < 1e-9


def test_many_chains_each_is_a_reference_chain():
    from klhr_amd import KLHRSINH
    algo = KLHRSINH(funnel_model(), seed=0, warmup=100, overrelaxed=False, chains=4)     # chain c <- default_rng(0 + c)
    dr = algo.draws(50)
    assert dr.shape == (50, 4, 10)
    for c in range(4):
        g = golden(f"g8_funnel10_s{c}_w100.npz")
        assert np.abs(dr[:, c] - g["theta_after"][:50]).max() < 1e-9, c
    assert algo.acceptance_probability.shape == (4,) and algo.grad_evals.shape == (4,)


def test_pieces_with_reference_method_names(oracle):
    from klhr_amd import KLHRSINH
    g = golden("g8_funnel10_s1_w100.npz")
    algo = KLHRSINH(funnel_model(), seed=1, warmup=100, overrelaxed=False)
    rho = algo._random_direction()
    assert np.array_equal(rho, g["rho"][0])
    eta = algo.fit(rho)
    assert np.allclose(eta, g["eta"][0], rtol=1e-6, atol=1e-8)
    v, grad = algo.KL(eta, rho)
    assert np.max(np.abs(grad)) < 1e-3          # BFGS stopped at gtol = 1e-3 (klhr_sinh.py:199)
    f, df = algo.logp_grad_rho(g["mode"][0], rho)
    assert abs(df) < 1e-5
    th = algo._metropolis_step(g["eta"][0], rho)
    assert np.allclose(th, g["theta_after"][0], rtol=1e-10, atol=1e-12)
    assert algo._log_q(0.3, eta) == pytest.approx(float(oracle.Chain("funnel", {"D": 9}).primitives_inv(0.3, eta)["log_q"]), rel=1e-12)
    assert algo._grad_T(0.0, eta).shape == (4,)


def test_reference_constructor_defaults_work():
    """KLHRSINH(model) with the reference's constructor defaults (overrelaxed=True, warmup=1000: klhr_sinh.py:15-32)."""
    from klhr_amd import BSModel, KLHRSINH
    m = BSModel(os.path.join(STAN, "normal.stan"), os.path.join(STAN, "normal.json"))
    algo = KLHRSINH(m, seed=4, chains=512)
    dr = algo.draws(1500)[300:].reshape(-1, 2)
    assert np.abs(dr.mean(axis=0)).max() < 0.05 and np.abs(dr.var(axis=0) - 1).max() < 0.06
    assert 0.5 < algo.acceptance_probability.mean() <= 1.0


def test_mh_class_vs_golden():
    from klhr_amd import BSModel, MH
    g = golden("g9_mh.npz")
    m = BSModel(os.path.join(STAN, "normal.stan"), os.path.join(STAN, "normal.json"))
    mh = MH(m, 0.09, seed=0)
    assert np.array_equal(mh.theta, g["normal2_s0_theta0"])
    th = np.array([mh.draw() for _ in range(30)])
    assert np.allclose(th, g["normal2_s0_thetas"][:30], atol=1e-12)
    rest = mh.draws(1970)
    assert np.allclose(rest[-1], g["normal2_s0_last"], atol=1e-12)
    assert mh.acceptance_probability[0] == pytest.approx(g["normal2_s0_acc"][0], abs=1e-12)


def test_experiment_harnesses_run(tmp_path):
    from click.testing import CliRunner
    from klhr_amd.experiment_accuracy import main as acc_main
    from klhr_amd.experiment_funnel import main as funnel_main
    r = CliRunner().invoke(funnel_main, ["-M", "150", "-w", "100", "--chains", "8", "--seed", "1", "--outdir", str(tmp_path), "-v", "klhr_sinh"])
    assert r.exit_code == 0, r.output
    assert "Acceptance rate" in r.output and os.path.exists(tmp_path / "histogram_False_False_False_klhr_sinh.png")
    r = CliRunner().invoke(acc_main, ["-M", "300", "--seed", "2", "--outdir", str(tmp_path), "-v", "klhr_sinh"])
    assert r.exit_code == 0, r.output
    assert "final RMSE" in r.output
    # the reference's fourth algorithm choice (experiment_funnel.py:42-43, experiment_accuracy.py:56-63)
    r = CliRunner().invoke(funnel_main, ["-M", "150", "-w", "100", "--chains", "8", "--seed", "1", "--outdir", str(tmp_path), "--no-plots", "slice"])
    assert r.exit_code == 0 and "v = theta[0]" in r.output, r.output
    r = CliRunner().invoke(acc_main, ["-M", "300", "--seed", "2", "--outdir", str(tmp_path), "-v", "slice"])
    assert r.exit_code == 0 and "final RMSE" in r.output, r.output


def test_statistical_targets_funnel_and_normal():
    """The reference's own (eyeballed) integration checks as numbers: v ~ N(0, 3) on the funnel
    (experiment_funnel.py:66-72) and unit variances on stan/normal (experiment_accuracy.py:97)."""
    from klhr_amd import BSModel, KLHRSINH
    algo = KLHRSINH(funnel_model(), seed=11, warmup=100, overrelaxed=False, chains=1024)
    algo._sampler.step(3000)                     # v mixes slowly under this sampler: sd(v) 1.9 -> 2.8 over 4000 draws
    v = algo.draws(1000)[:, :, 0].reshape(-1)
    assert abs(v.mean()) < 0.4 and 2.4 < v.std() < 3.4, (v.mean(), v.std())
    m = BSModel(os.path.join(STAN, "normal.stan"), os.path.join(STAN, "normal-100.json"))
    algo = KLHRSINH(m, seed=5, warmup=100, overrelaxed=False, chains=256)
    dr = algo.draws(600)[200:].reshape(-1, 100)
    # hit-and-run moves along one direction per draw: in 100-D the chains are strongly autocorrelated
    assert np.abs(dr.mean(axis=0)).max() < 0.25 and np.abs(dr.var(axis=0) - 1).max() < 0.3


def test_pooled_adaptation_single_rank():
    from klhr_amd.parallel import ShardedKLHRSINH
    sh = ShardedKLHRSINH(funnel_model(), 64, seed=3, pooled=True, warmup=100)
    sh.advance(49)
    s = sh.local._sampler
    st = s.pool_stats()
    assert st[0] == 64 * 49
    om_m = s.get("om_m")
    assert np.allclose(st[1:11], 49 * om_m.sum(axis=0), rtol=1e-12)
    sh.advance(2)                       # crosses the closure at draw 50
    mean, cov = s.get("mean"), s.get("cov")
    assert np.all(mean == mean[0]) and np.all(cov == cov[0])         # every chain carries the pooled values
    assert np.allclose(mean[0], st[1:11] / st[0], rtol=1e-12)
    sh.advance(60)                      # and keeps sampling through the second closure
    assert np.isfinite(s.get("theta")).all()
    # parity mode leaves per-chain adaptation untouched
    sp = ShardedKLHRSINH(funnel_model(), 64, seed=3, pooled=False, warmup=100)
    sp.advance(51)
    m2 = sp.local._sampler.get("mean")
    assert not np.all(m2 == m2[0])


def test_mfma_gram_kernel_vs_numpy():
    """klhr_pool_gram: fp64 MFMA Gram product over recorded draws == numpy X^T X (plus column sums and the row count)."""
    from klhr_amd import _lib
    for D_alpha, C in ((9, 96), (99, 40)):
        s = _lib.Sampler(_lib.MODEL_IDS["funnel"], np.array([float(D_alpha)]), C, warmup=100)
        s.seed(np.arange(C, dtype=np.uint64) + 5)
        n = 7
        s.step(n, record_draws=True)
        X = s.draws(0, n).reshape(-1, s.D)
        g = s.pool_gram(0, n)
        D = s.D
        assert g[0] == n * C
        assert np.allclose(g[1:1 + D], X.sum(axis=0), rtol=1e-12, atol=1e-12)
        G = g[1 + D:].reshape(D, D)
        assert np.allclose(G, X.T @ X, rtol=1e-12, atol=1e-10)
        g2 = s.pool_gram(2, 3)                       # a sub-range of the buffer
        X2 = s.draws(2, 3).reshape(-1, D)
        assert np.allclose(g2[1 + D:].reshape(D, D), X2.T @ X2, rtol=1e-12, atol=1e-10) and g2[0] == 3 * C


def test_pooled_full_covariance_single_rank():
    from klhr_amd.parallel import ShardedKLHRSINH, pooled_full
    sh = ShardedKLHRSINH(funnel_model(), 256, seed=9, pooled="full", warmup=100)
    dr = sh.advance(49, record=True)
    s = sh.local._sampler
    X = dr.reshape(-1, 10)
    s.pool_allreduce()                 # (no communicator: packs the device-resident statistics)
    acc = s.pool_fetch()
    assert acc[0] == X.shape[0] and np.allclose(acc[1 + 10:].reshape(10, 10), X.T @ X, rtol=1e-11, atol=1e-9)
    sh.advance(2)                      # crosses the closure at draw 50 -> pooled mean / cov / eigvecs on every chain
    mean, cov, evec, evals = s.get("mean"), s.get("cov"), s.get("eigvecs"), s.get("eigvals")
    assert np.all(mean == mean[0]) and np.all(evec == evec[0]) and np.all(evals == evals[0])
    S = np.cov(X.T)
    w, V = np.linalg.eigh(S)
    assert np.allclose(mean[0], X.mean(axis=0), rtol=1e-10)
    assert np.allclose(cov[0], np.diag(S), rtol=1e-9)
    assert np.allclose(evals[0], w[::-1][:2], rtol=1e-8)
    assert abs(abs(evec[0][:, 0] @ V[:, -1]) - 1) < 1e-8
    sh.advance(60)                     # keeps sampling through the second closure with pooled directions
    assert np.isfinite(s.get("theta")).all()


def test_pooled_full_through_rccl_single_rank_equals_host_path():
    """The collective inside the library: klhr_comm_init with a one-rank communicator, klhr_pool_allreduce = ncclAllReduce on
    the handle's stream.  The pooled values installed on the chains must equal those of the host (no-communicator) path."""
    from klhr_amd.parallel import ShardedKLHRSINH
    res = []
    for use_rccl in (False, True):
        sh = ShardedKLHRSINH(funnel_model(), 192, seed=11, pooled="full", rccl=use_rccl, warmup=100)
        s = sh.local._sampler
        assert s.comm_world() == (1 if use_rccl else 0)
        sh.advance(50)                 # 49 accumulated draws, all-reduce, closure iteration, pooled values applied
        res.append((s.get("mean"), s.get("cov"), s.get("eigvecs"), s.get("eigvals"), s.get("theta")))
        sh.advance(55)                 # through the second closure
        assert np.isfinite(s.get("theta")).all()
    # (the Gram kernel combines its slabs with floating-point atomics: two runs agree to rounding, not bit for bit)
    for a, b in zip(res[0][:4], res[1][:4]):
        assert np.allclose(a, b, rtol=1e-9, atol=1e-12)
    assert np.array_equal(res[0][4], res[1][4])        # theta after the closure draw does not depend on the pooled values yet


def test_lds_tiled_gram_matches_numpy_over_dimensions():
    """klhr_pool_accumulate (LDS-tiled MFMA Gram, X read once) for every tile count T = 1..8 and the wide fallback, with row
    counts that are not multiples of the chunk, accumulated over two segments."""
    from klhr_amd import _lib
    for D, C in ((3, 50), (20, 33), (40, 70), (50, 64), (70, 17), (90, 40), (100, 129), (120, 9), (140, 12)):
        s = _lib.Sampler(_lib.MODEL_IDS["normal"], np.array([float(D)]), C, warmup=100)
        s.seed(np.arange(C, dtype=np.uint64) + 77)
        s.step(5, record_draws=True)
        X1 = s.draws(0, 5).reshape(-1, D)
        s.pool_reset()
        s.pool_accumulate(0, 5)
        s.step(3, record_draws=True)
        X2 = s.draws(1, 2).reshape(-1, D)
        s.pool_accumulate(1, 2)
        s.pool_allreduce()
        g = s.pool_fetch()
        X = np.concatenate([X1, X2])
        assert g[0] == X.shape[0], (D, C)
        assert np.allclose(g[1:1 + D], X.sum(axis=0), rtol=1e-12, atol=1e-11), (D, C)
        assert np.allclose(g[1 + D:].reshape(D, D), X.T @ X, rtol=1e-12, atol=1e-10), (D, C)


def test_klhr_class_is_the_reference_chain():
    from klhr_amd import KLHR
    g = golden("g11_klhr_public_api.npz")
    algo = KLHR(funnel_model(), seed=0, warmup=100)        # klhr.py defaults: eigen_method_one=True, overrelaxed=False
    th = np.array([algo.draw() for _ in range(50)])
    assert np.abs(th - g["s0_thetas"][:50]).max() < 1e-9
    more = algo.draws(70)
    assert algo.K == int(g["s0_K"][0])
    assert np.isfinite(more).all() and algo.acceptance_probability.shape == (1,)
    rho = algo._random_direction()
    eta = algo.fit(rho)
    assert eta.shape == (2,)
    v, grad = algo.KL(eta, rho)
    assert np.max(np.abs(grad)) < 1e-4                     # stage-2 BFGS runs to the default gtol = 1e-5 (klhr.py:136-139)
    many = KLHR(funnel_model(), seed=3, warmup=100, chains=300, overrelaxed=True)
    many.draws(150)
    assert many.K.shape == (300,) and many.K.min() >= 1 and many.K.max() <= 50


def test_subklhrsinh_class():
    from klhr_amd import SUBKLHRSINH
    g = golden("g12_sub_public_api.npz")
    algo = SUBKLHRSINH(funnel_model(), seed=0, warmup=100, overrelaxed=False)
    th = np.array([algo.draw() for _ in range(50)])
    assert np.abs(th - g["s0_thetas"][:50]).max() < 1e-9
    rho = algo._random_direction()
    eta = algo.fit(rho)
    assert eta.shape == (3,)
    v, grad = algo.KL(eta, rho)
    assert grad.shape == (3,) and np.max(np.abs(grad)) < 1e-3
    assert algo._grad_T(0.2, eta).shape == (3,)
    many = SUBKLHRSINH(funnel_model(), seed=1, warmup=100, chains=128)      # reference default: overrelaxed=True
    assert np.isfinite(many.draws(120)).all()


def test_ar1_and_earnings_targets_and_harnesses(tmp_path):
    """the ar1 / earnings targets with their experiment counterparts (experiment_ar1.py:96-117,
    experiment_relaxationtime.py:80-106)."""
    from click.testing import CliRunner
    from klhr_amd import BSModel, KLHRSINH
    from klhr_amd.experiment_ar1 import main as ar1_main
    from klhr_amd.experiment_relaxationtime import main as relax_main
    g = golden("g8_ar1x100_s0_w100.npz")
    m = BSModel(os.path.join(STAN, "ar1.stan"), os.path.join(STAN, "ar1.json"))
    algo = KLHRSINH(m, seed=0, warmup=100, overrelaxed=False)
    th = np.array([algo.draw() for _ in range(40)])
    # (free-running: each draw's 1e-6-level eta difference feeds the next state, and on this Gaussian target the
    #  flat KL makes that amplification faster than on the funnel — tight early, bounded over the 40)
    assert np.abs(th[:10] - g["theta_after"][:10]).max() < 1e-8
    assert np.abs(th - g["theta_after"][:40]).max() < 1e-5
    ge = golden("g8_earnings_s0_w100.npz")
    me = BSModel(os.path.join(STAN, "earnings.stan"), os.path.join(STAN, "earnings.json"))
    assert me.dim() == 4 and me.parameter_names() == ["beta.1", "beta.2", "sigma", "s"]
    ae = KLHRSINH(me, seed=0, warmup=100, overrelaxed=False)
    the = np.array([ae.draw() for _ in range(30)])
    # (most earnings fits end in a failed line search — status 2 from the first record on — so the fitted eta is
    #  rounding-noise dependent, cf. test_teacher_forced_pieces_vs_oracle: free-running agreement is short-lived)
    assert np.allclose(the[:2], ge["theta_after"][:2], rtol=1e-4, atol=1e-4)
    assert np.isfinite(the).all()
    r = CliRunner().invoke(ar1_main, ["-M", "300", "-w", "100", "--chains", "64", "--seed", "1", "--outdir", str(tmp_path), "-v", "klhr_sinh"])
    assert r.exit_code == 0 and "RMSE(mean)" in r.output, r.output
    r = CliRunner().invoke(relax_main, ["-M", "200", "-w", "100", "--seed", "2", "--outdir", str(tmp_path), "-v", "klhr"])
    assert r.exit_code == 0 and "Number gradients" in r.output, r.output


TWO_RANK_WORKER = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, os.environ["KLHR_ROOT"])
import torch
import torch.distributed as dist
from klhr_amd import BSModel
from klhr_amd.parallel import ShardedKLHRSINH, dist_env
rank, world, _ = dist_env()
dist.init_process_group(backend="gloo", rank=rank, world_size=world)
def host_sum(vec):          # two ranks share the one GPU: RCCL cannot form a communicator, the caller injects the stand-in
    t = torch.from_numpy(np.array(vec, dtype=np.float64))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.numpy()
stan = os.path.join(os.environ["KLHR_ROOT"], "klhr_amd", "stan")
m = BSModel(os.path.join(stan, "funnel.stan"), os.path.join(stan, "funnel.json"))
out = {}
for mode in (False, "diag", "full"):
    sh = ShardedKLHRSINH(m, 96, seed=5, pooled=mode, dist=dist, rccl=False, host_allreduce=host_sum, warmup=100)
    sh.advance(50)                                   # up to and including the first window closure
    s = sh.local._sampler
    first = {"mean50": s.get("mean").tolist(), "cov50": s.get("cov").tolist()}
    dr = sh.advance(70, record=True)
    out[str(mode)] = {**first, "lo": sh.lo, "hi": sh.hi, "theta": s.get("theta").tolist(), "mean": s.get("mean").tolist(),
                      "cov": s.get("cov").tolist(), "eigvecs": s.get("eigvecs").tolist(), "draws_shape": list(dr.shape)}
dist.barrier()
with open(os.path.join(os.environ["KLHR_OUT"], f"gpu_result_{rank}.json"), "w") as f:
    json.dump(out, f)
dist.destroy_process_group()
'''


def test_two_ranks_on_the_device_match_one_rank(tmp_path):
    """The multi-GPU layout end to end (two processes, here sharing the one GPU; gloo for the collective): parity mode
    reproduces the single-rank chains bit for bit; the pooled modes install the same pooled mean / covariance / directions
    on every chain of every rank, equal to the single-rank pooled values up to the order of the cross-rank sum."""
    import json, subprocess, sys
    from klhr_amd.parallel import ShardedKLHRSINH
    script = tmp_path / "worker.py"
    script.write_text(TWO_RANK_WORKER)
    env = dict(os.environ, KLHR_ROOT=ROOT, KLHR_OUT=str(tmp_path), MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29741", str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    res = [json.load(open(tmp_path / f"gpu_result_{r}.json")) for r in range(2)]
    for mode in (False, "diag", "full"):
        one = ShardedKLHRSINH(funnel_model(), 96, seed=5, pooled=mode, warmup=100)
        one.advance(50)
        s1 = one.local._sampler
        mean50, cov50 = s1.get("mean").copy(), s1.get("cov").copy()
        one.advance(70)
        both = {k: np.concatenate([np.asarray(res[r][str(mode)][k]) for r in range(2)]) for k in ("theta", "mean", "cov", "eigvecs", "mean50", "cov50")}
        assert (res[0][str(mode)]["lo"], res[0][str(mode)]["hi"], res[1][str(mode)]["lo"], res[1][str(mode)]["hi"]) == (0, 48, 48, 96)
        assert res[0][str(mode)]["draws_shape"] == [70, 48, 10]
        if mode is False:
            for k in ("theta", "mean", "cov", "eigvecs"):
                assert np.array_equal(both[k], s1.get(k)), k
        else:
            assert np.all(both["mean"] == both["mean"][0]) and np.all(both["cov"] == both["cov"][0])
            # pooled over 96 chains by two partial sums vs one sum: equal to rounding at the first closure (afterwards
            # the chains evolve from pooled values that differ in the last bits, so later states are not compared)
            assert np.allclose(both["mean50"], mean50, rtol=1e-11, atol=1e-13)
            assert np.allclose(both["cov50"], cov50, rtol=1e-9, atol=1e-13)
            assert np.isfinite(both["theta"]).all()


def test_one_normal_like_slice_main():
    """slice.py:185-208 (`__main__`): Slice on stan/one_normal (a 1-D standard normal), 10 000 draws, mean and sd of the
    draws; the same 1-D target through KLHRSINH."""
    from klhr_amd import BSModel, KLHRSINH, Slice
    m = BSModel(os.path.join(STAN, "one_normal.stan"), os.path.join(STAN, "one_normal.json"))
    assert m.dim() == 1
    draws = Slice(m, seed=4).sample(10_000)
    assert draws.shape == (10_000, 1) and abs(np.mean(draws)) < 0.06 and abs(np.std(draws, ddof=1) - 1.0) < 0.05
    ks = KLHRSINH(m, seed=4, warmup=100, overrelaxed=False, chains=256)
    assert np.all(ks._sampler.get("eigvals") == 1.0)            # J = 2 eigenvalue slots although D = 1 (klhr_sinh.py:66)
    many = ks.draws(300)[120:]
    assert abs(many.mean()) < 0.03 and abs(many.std() - 1.0) < 0.04
    # ncol = J + 1 > D (eigen_method_one on a 2-D target): the extra eigenvalue slot starts at 1 and is never overwritten by a
    # window closure (klhr_sinh.py:64-66,276-277), exactly as in the oracle
    from oracle import oracle as O
    m2 = BSModel(os.path.join(STAN, "normal.stan"), os.path.join(STAN, "normal.json"))
    k2 = KLHRSINH(m2, seed=7, warmup=100, overrelaxed=False, eigen_method_one=True)
    c2 = O.Chain("normal", {"D": 2}, seed=7, warmup=100, eigen_method_one=True)
    assert np.array_equal(k2._sampler.get("eigvals")[0], [1.0, 1.0, 1.0])
    th = np.array([k2.draw() for _ in range(51)])
    tho = c2.draw_many(51)
    assert np.allclose(th[:50], tho[:50], rtol=1e-8, atol=1e-9)
    ev, evo = k2._sampler.get("eigvals")[0], c2.get("eigvals")
    assert ev[2] == 1.0 and evo[2] == 1.0 and np.allclose(ev[:1], evo[:1], rtol=1e-6)


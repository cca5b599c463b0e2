"""Helpers shared by the GPU parity tests: run the CPU oracle teacher-forced over a golden chain trace and
collect, per record, the state before the draw and the oracle's output of every piece of draw()."""
import numpy as np

from conftest import golden, model_spec

ADAPT_FIELDS = ["mean", "cov", "eigvecs", "eigvals", "om_m", "om_v", "pca_v"]


def parse_g8(fn):
    parts = fn[3:-4].split("_")
    flags = {}
    if parts[-1] == "e":
        flags["eigen_method_one"] = True
    if parts[-1] == "s":
        flags["scale_dir_cov"] = True
    return parts[0], flags


def oracle_teacher_forced(oracle, fn, max_records=None):
    name, flags = parse_g8(fn)
    g = golden(fn)
    stem, data = model_spec(name)
    M, warmup, seed = int(g["M"][0]), int(g["warmup"][0]), int(g["seed"][0])
    if max_records:
        M = min(M, max_records)
    c = oracle.Chain(stem, data, seed=seed, warmup=warmup, **flags)
    fields = ADAPT_FIELDS + (["omd_m", "omd_v"] if flags.get("scale_dir_cov") else [])
    before = {f: [] for f in fields}
    after = {f: [] for f in fields}
    cnt_before, cnt_after = [], []
    out = dict(rho=[], eta=[], diag_fit=[], rng_after_dir=[], rng_after_fit=[], theta_mh=[], diag_mh=[], rng_after_mh=[],
               acc_before=[], acc_after=[], gev_before=[], gev_after=[])
    for m in range(M):
        for f in fields:
            before[f].append(c.get(f).copy())
        cnt_before.append([m, c.get("om_N"), c.get("pca_n"), c.get("cidx")])
        c.set("theta", g["theta_before"][m])
        c.set("rng", g["state_before"][m])
        out["rho"].append(c.random_direction())
        out["rng_after_dir"].append(c.get("rng"))
        out["gev_before"].append(c.get("grad_evals"))
        out["eta"].append(c.fit(g["rho"][m]))
        out["gev_after"].append(c.get("grad_evals"))
        out["diag_fit"].append(c.get("diag"))
        out["rng_after_fit"].append(c.get("rng"))
        c.set("draw", m)
        out["acc_before"].append(c.get("acc"))
        c.metropolis(g["eta"][m], g["rho"][m])
        out["acc_after"].append(c.get("acc"))
        out["theta_mh"].append(c.get("theta"))
        out["diag_mh"].append(c.get("diag"))
        out["rng_after_mh"].append(c.get("rng"))
        c.set("theta", g["theta_after"][m])
        c.adapt()
        for f in fields:
            after[f].append(c.get(f).copy())
        cnt_after.append([m + 1, c.get("om_N"), c.get("pca_n"), c.get("cidx")])
    res = dict(name=name, flags=flags, stem=stem, data=data, M=M, warmup=warmup, seed=seed, g=g, fields=fields,
               before={k: np.array(v) for k, v in before.items()}, after={k: np.array(v) for k, v in after.items()},
               cnt_before=np.array(cnt_before, dtype=np.int64), cnt_after=np.array(cnt_after, dtype=np.int64))
    for k, v in out.items():
        res[k] = v if k.startswith("diag") else np.array(v)
    return res


def make_sampler(stem, data, chains, **kw):
    from klhr_amd import _lib
    from oracle.oracle import model_data
    return _lib.Sampler(_lib.MODEL_IDS[stem], model_data(stem, data), chains, **kw)


STATE_FIELDS = ["theta", "rng", "mean", "cov", "eigvecs", "eigvals", "om_m", "om_v", "pca_v"]


def oracle_state_trace(oracle, stem, data, seed, n, **kw):
    """Run one oracle chain for n draws; per draw the chain's full state BEFORE it (every field a device chain needs to take the same
    draw: theta, generator, adaptation state, counters) and theta / generator / acceptance count AFTER it.  Lets a test hand
    every draw of a trajectory to its own device chain (teacher forcing with the oracle's own states, any constructor arguments)."""
    c = oracle.Chain(stem, data, seed=seed, **kw)
    before = {f: [] for f in STATE_FIELDS}
    cnt, acc_b, theta_a, rng_a, acc_a, eta_a, diag_a = [], [], [], [], [], [], []
    for m in range(n):
        for f in STATE_FIELDS:
            before[f].append(np.array(c.get(f), copy=True))
        cnt.append([c.get("draw"), c.get("om_N"), c.get("pca_n"), c.get("cidx")])
        acc_b.append(c.get("acc"))
        c.draw()
        theta_a.append(c.get("theta").copy()); rng_a.append(c.get("rng").copy()); acc_a.append(c.get("acc"))
        eta_a.append(c.get("eta").copy()); diag_a.append(c.get("diag"))
    out = {f: np.array(v) for f, v in before.items()}
    out.update(counters=np.array(cnt, dtype=np.int64), acc_before=np.array(acc_b), theta_after=np.array(theta_a), rng_after=np.array(rng_a),
               acc_after=np.array(acc_a), eta_after=np.array(eta_a), diag_after=diag_a)
    return out

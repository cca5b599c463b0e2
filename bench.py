#!/usr/bin/env python3
"""bench.py — headline benchmark: MCMC iterations/sec (all chains) + ESS/sec on the funnel-10D target.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

A "step" is one KLHRSINH.draw() for every chain of the workload (BASELINE.json configs[1]: funnel 10-dim, 65 536
chains per MI355X, log density as a HIP device function).  Chains shard across GPUs with no data-path collective
(weak scaling: 65 536 chains per GPU); torch.distributed (gloo, CPU tensors) is used only for the barrier and the
max-over-ranks timing the contract asks for.  `--pooled full` adds the north-star's pooled adaptation (one RCCL
all-reduce of 1 + D + D^2 doubles per window closure, inside the library) to the untimed setup.

What is timed.  Setup (untimed, independent of --warmup): the sampler's own adaptation — its warmup=100 draws, so both
adaptation windows have closed (experiment_funnel.py:21 default) — plus 50 more draws.  Then W untimed warm-up steps, then
REPEATS (5) regions of exactly K steps, each bracketed by barrier + device sync on both sides, max over ranks; the
MEDIAN region gives `ms_per_step` and `value`, min / max are reported beside it.  Inputs (chain state) are resident in HBM
when a timed region starts; draws are recorded into an HBM buffer inside it and never copied to the host there.
ESS/sec comes from a separate, fixed run outside the timed regions (512 chains: setup, 3 000 burn-in draws, 20 000 recorded
draws): ESS per chain-iteration x the measured iterations/sec, reported only when Geyer's truncation lag is below a tenth of
the run (otherwise null + reason).  With N > 1 ranks every rank also runs BASELINE config 5's per-GPU shard (funnel-100D,
131 072 chains) with pooled="full" adaptation — the RCCL all-reduce inside the library — and rank 0 reports it under
`configs` and `rccl`; the headline stays parity mode (no collective).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6   # MI355X fp64 vector peak (half of the 157.3 TF fp32 vector rate)
SETUP_DRAWS = 150              # untimed: the sampler's warmup = 100 draws (both windows close) + 50
REPEATS = 5
POOLED_TIMEOUT_S = 240           # N > 1: the pooled shard (RCCL) gets this long before the watchdog reports and gives up


def algorithmic_bytes_per_iter(D, J=2):
    """reads (5+2J)D+J doubles, writes (4+J)D doubles (draw recorded), + RNG state and scalars."""
    return 8 * ((5 + 2 * J) * D + J + (4 + J) * D) + 112


def flop_equiv_per_iter(D, grad_evals=None):
    """~14 KL evals x 8 nodes + ~8.5 single evals of (9D+10 flops + 1 transcendental) + ~40 flops and 3
    transcendentals per node + ~11 BFGS updates of ~250 flops; a transcendental counted as 25 flops.  With a measured
    evaluation count the node-evaluation terms are scaled to it (glmm-poisson needs ~4x the evaluations of the funnel)."""
    node_evals = 14.0 * 8 + 8.5
    scale = (grad_evals / node_evals) if grad_evals else 1.0
    return scale * (node_evals * (9 * D + 10 + 25) + 14.0 * 8 * (40 + 3 * 25) + 11 * 250)


def model_data(model, dim):
    if model == "funnel":
        return np.array([float(dim - 1)]), dim
    if model == "glmm-poisson":
        d = json.load(open(os.path.join(ROOT, "klhr_amd", "stan", "glmm-poisson.json")))
        return np.concatenate([[float(d["n"])], d["year"], d["C"]]), d["n"] + 5
    return np.array([float(dim)]), dim


def cpu_model_string():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline_worker(chains, warm, iters):
    """(child process, OMP_NUM_THREADS set by the parent) the CPU oracle timed on host cores; prints iterations/s"""
    from oracle import oracle as O
    O.build()
    cs = [O.Chain("funnel", {"D": 9}, seed=10_000 + i, warmup=100) for i in range(chains)]
    O.draw_batch(cs, warm)
    t0 = time.perf_counter()
    O.draw_batch(cs, iters)
    dt = time.perf_counter() - t0
    print(chains * iters / dt)


def cpu_baseline(chains, warm, iters, threads):
    """The CPU oracle (oracle/klhr_oracle.c, a validated restatement of the reference path) timed on `threads` host cores.
    Runs in a child process: OpenMP reads OMP_NUM_THREADS once, when the oracle library is first loaded."""
    import subprocess
    env = dict(os.environ, OMP_NUM_THREADS=str(threads))
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", str(chains), str(warm), str(iters)],
                       env=env, capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        raise RuntimeError("cpu baseline worker failed: " + r.stderr[-500:])
    return float(r.stdout.strip().splitlines()[-1])


def traffic_entry(key):
    """profiles/traffic.json[key] if it was measured on the library build that is loaded (klhr_build_id == the entry's
    build_id), else (None, reason): counters of a kernel that has changed since are not quoted."""
    from klhr_amd import _lib
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        te = json.load(open(tpath)).get(key)
    except Exception as e:
        return None, f"profiles/traffic.json unreadable: {e}"
    if te is None:
        return None, f"no entry {key!r} in profiles/traffic.json"
    have = _lib.build_id()
    if te.get("build_id") != have:
        return None, (f"profiles/traffic.json[{key!r}] was measured on build {str(te.get('build_id'))[:16]}, the loaded library is "
                      f"{have[:16]}: re-run tools/profile_cfg.sh + tools/profile_cfg_collect.py")
    return te, None


def run_workload(model, dim, C, steps, warm, repeats, rank, world, device, seed, dist, pooled=None, profile_iters=20):
    """Set up one sampler, run `repeats` timed regions of `steps` draws, then a short per-kernel profiling pass."""
    from klhr_amd import _lib
    data, D = model_data(model, dim)
    s = _lib.Sampler(_lib.MODEL_IDS[model], data, C, device=device, warmup=100)   # experiment_funnel.py:21 default
    seeds = np.uint64(seed) + np.uint64(rank) * np.uint64(C) + np.arange(C, dtype=np.uint64)
    s.seed(seeds)

    def barrier():
        if dist is not None:
            dist.barrier()

    comm = None
    if pooled:
        from klhr_amd.parallel import pooled_setup_draws

        def host_sum(vec):          # stand-in for rehearsals whose ranks share one GPU (RCCL needs one device per rank)
            import torch
            t = torch.from_numpy(np.array(vec, dtype=np.float64))
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            return t.numpy()
        comm = pooled_setup_draws(s, SETUP_DRAWS, rank, world, dist, mode=pooled, host_allreduce=host_sum if dist is not None else None)
    else:
        s.step(SETUP_DRAWS)                 # the sampler's own adaptation: untimed setup, whatever --warmup says
    if warm > 0:
        s.step(warm)
    s.reserve_draws(steps)                  # the output buffer (np.zeros((M, D)) in experiment_funnel.py:48) exists before the clock starts
    region, region_kernel = [], []
    for _ in range(repeats):
        s.sync()
        barrier()
        s.sync()
        t0 = time.perf_counter()
        s.step(steps, record_draws=True, sync=False)
        s.sync()
        barrier()
        dt = time.perf_counter() - t0
        region_kernel.append(s.timing())     # HIP events around this region's launches on the handle's stream: (ms, launches)
        if dist is not None:
            import torch
            t = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t[0])
        region.append(dt)
    dt_med = float(np.median(region))
    total_iters = world * C * steps
    value = total_iters / dt_med
    names = s.kernel_names()
    mono = len(names) == 1
    b_iter = algorithmic_bytes_per_iter(D)
    kernels = {}
    if mono:
        # one hot kernel: its launch duration comes from the TIMED regions' own HIP events (median region), so the kernel time
        # per step can never exceed ms_per_step
        kms = sorted(ms / max(nl, 1) for ms, nl in region_kernel)[len(region_kernel) // 2]
        nl = region_kernel[0][1]
        kernels[names[0]] = {"mean_ms": kms, "launches": nl * repeats}
        iters_per_launch = steps / max(nl, 1)
        times_from = "HIP events around the timed regions' launches (median region)"
    else:
        # split pipeline: the timed regions run 2-4 chain groups on separate streams (kernels of different groups overlap), so
        # per-kernel durations come from a separate pass with ONE group, every kernel bracketed by HIP events on its stream
        s.kernel_profiling(True)
        n_prof = min(steps, profile_iters)
        s.step(n_prof, record_draws=True)
        kms3, kn3 = s.kernel_times()
        s.kernel_profiling(False)
        for nm, ms, n in zip(names, kms3, kn3):
            if n:
                kernels[nm] = {"mean_ms": ms, "launches": n}
        iters_per_launch = 1.0
        times_from = f"separate untimed pass of {n_prof} draws with one chain group (the timed regions overlap the groups' kernels)"
    ge = s.get("grad_evals").astype(np.float64)
    cnt = s.get("counters")[:, 0].astype(np.float64)
    grad_evals = float((ge / np.maximum(cnt, 1)).mean())
    # the roofline's kernel: the one that moves the iteration's algorithmic bytes — the hot kernel of the monolithic mappings, the
    # streaming kernel of the split pipeline (its fit kernels work on a few doubles per chain; `longest_kernel` says which one
    # took longest, `whole_iteration` is the figure over all kernels of an iteration)
    longest = max(kernels, key=lambda k: kernels[k]["mean_ms"] * kernels[k]["launches"])
    dom = names[0] if names[0] in kernels else longest
    launch_s = kernels[dom]["mean_ms"] / 1e3
    achieved_gbs = b_iter * C * iters_per_launch / launch_s / 1e9
    tkey = f"{model}-{D}-{C}"
    te, stale = traffic_entry(tkey)
    traffic = te["hbm_bytes_per_launch"] * (iters_per_launch / te["iters_per_launch"]) if te else None
    fe = flop_equiv_per_iter(D, grad_evals)
    fp64 = {"algorithmic": {"achieved_tflops": value / world * fe / 1e12, "frac": value / world * fe / 1e12 / FP64_VALU_PEAK_TFLOPS,
                            "flop_equivalents_per_chain_iter": fe,
                            "note": "SURVEY 8d's flop-equivalents of the element-wise O(D) evaluation x measured it/s; the closed-form kernels do not execute them"},
            "peak_tflops": FP64_VALU_PEAK_TFLOPS}
    if te and te.get("fp64_flops_per_launch"):
        # counter-derived: (2 FMA + ADD + MUL + TRANS) fp64 instructions x 64 lanes x the lanes-active share, per launch of the
        # dominant kernel (profiles/<tag>/pmc_sq.csv), over this run's launch duration
        mf = te["fp64_flops_per_launch"] * (iters_per_launch / te["iters_per_launch"]) / launch_s / 1e12
        fp64["measured"] = {"achieved_tflops": mf, "frac": mf / FP64_VALU_PEAK_TFLOPS, "source": te.get("fp64_source"),
                            "valu_insts_per_wavefront_iteration": te.get("valu_per_wavefront_iter"), "lanes_active": te.get("lanes_active"),
                            "valu_active": te.get("valu_active")}
    else:
        fp64["measured"] = None
    out = {
        "value": value, "ms_per_step": dt_med * 1e3 / steps,
        "ms_per_step_min": min(region) * 1e3 / steps, "ms_per_step_max": max(region) * 1e3 / steps, "timed_regions": repeats,
        "roofline": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved_gbs / HBM_PEAK_GBS,
                     "traffic": traffic, "kernel": dom, "longest_kernel": longest, "algorithmic_bytes_per_chain_iter": b_iter,
                     "launch_ms": launch_s * 1e3, "iters_per_launch": iters_per_launch, "kernels": kernels, "kernel_times_from": times_from,
                     "whole_iteration": {"achieved": b_iter * value / world / 1e9, "frac": b_iter * value / world / 1e9 / HBM_PEAK_GBS,
                                         "note": "algorithmic bytes x measured iterations/s (all kernels of an iteration, per GPU)"},
                     "fp64_valu": fp64},
        "grad_evals_per_iter": grad_evals, "acceptance": float(s.get("acc").mean()), "kernel": s.kernel_name(),
    }
    if traffic is not None:
        out["roofline"]["traffic_source"] = te.get("source")
        out["roofline"]["traffic_over_algorithmic"] = traffic / (b_iter * C * iters_per_launch)
    else:
        out["roofline"]["traffic_stale"] = True
        out["roofline"]["traffic_reason"] = stale
    if comm is not None:
        out["pooled"] = comm
    s.close()
    return out, D


def ess_run(device, seed, value):
    """ESS of the funnel-10D chains from a fixed run outside every timed region.  The funnel's v mixes slowly under hit-and-run
    (measured: integrated autocorrelation time ~5 000 draws), so the run is long and narrow: 128 chains, the same setup draws,
    20 000 burn-in draws, then 250 000 recorded draws (~45 s: a launch of two wavefronts is as long as its chain of dependent
    iterations, ~0.17 ms each, however few chains it holds).  ESS/sec = ESS per chain-iteration x the headline iterations/sec
    (all chains).  An estimate whose Geyer truncation lag is not below a tenth of the run is bounded by the window rather than
    by the chain: reported as null with the reason; `tau.between_chains` (run length x variance of the chain means / pooled
    variance) is the cross-check that needs no truncation."""
    from klhr_amd import _lib
    from klhr_amd.ess import ess_detail
    nch, burn, ndraw = 128, 20000, 250000
    data, D = model_data("funnel", 10)
    s = _lib.Sampler(_lib.MODEL_IDS["funnel"], data, nch, device=device, warmup=100)
    s.seed(np.uint64(seed) + np.uint64(7_000_000) + np.arange(nch, dtype=np.uint64))
    s.step(SETUP_DRAWS)
    s.step(burn)
    s.sync()
    t0 = time.perf_counter()
    s.step(ndraw, record_draws=True)
    dt = time.perf_counter() - t0
    dr = s.draws(0, ndraw)
    s.close()
    det = [ess_detail(dr[:, :, j]) for j in range(D)]
    v = dr[:, :, 0]
    tau_between = float(ndraw * v.mean(axis=0).var(ddof=1) / v.var())

    def per_iter(d):
        return d["ess"] / (nch * ndraw)
    jmin = int(np.argmin([d["ess"] for d in det]))
    half = [ess_detail(dr[: ndraw // 2, :, j])["ess"] / (nch * (ndraw // 2)) for j in (0, jmin)]      # stability: the same on half the run
    # when is the estimate quoted?  (i) Geyer's truncation lag below a tenth of the run for every coordinate, or (ii) for the slowest
    # coordinate two independent estimators agree: Geyer's tau from the autocorrelations and the between-chain tau (run length x
    # variance of the 128 chain means / pooled variance, which needs no truncation; relative error sqrt(2/127) = 13 %) within 20 %,
    # on a run of at least 20 tau.  Otherwise null + reason.
    lag_ok = all(d["truncation_lag"] < ndraw / 10 for d in det)
    agree = abs(det[0]["tau"] / tau_between - 1.0) < 0.2 and ndraw >= 20 * max(det[0]["tau"], tau_between)
    ok = lag_ok or agree
    out = {"v": per_iter(det[0]) * value if ok else None, "min_coord": per_iter(det[jmin]) * value if ok else None,
           "v_from_between_chain_tau": value / tau_between,
           "quoted_because": ("Geyer truncation lag < draws / 10" if lag_ok else
                              ("Geyer tau and between-chain tau of v agree within 20 % on a run of >= 20 tau" if agree else None)),
           "ess_per_chain_iteration": {"v": per_iter(det[0]), "min_coord": per_iter(det[jmin]), "v_on_half_the_run": half[0],
                                       "min_coord_on_half_the_run": half[1]},
           "tau": {"v": det[0]["tau"], "min_coord": det[jmin]["tau"], "max_truncation_lag": max(d["truncation_lag"] for d in det),
                   "draws_per_chain": ndraw, "v_between_chains": tau_between},
           "rhat": {"v": det[0]["rhat"], "max": max(d["rhat"] for d in det)},
           "moments_of_v": {"mean": float(dr[:, :, 0].mean()), "sd": float(dr[:, :, 0].std()), "target": "N(0, 3)"},
           "estimated_on": {"chains": nch, "burn_in_draws": SETUP_DRAWS + burn, "draws": ndraw, "its_own_it_per_s": nch * ndraw / dt},
           "method": "multi-chain ESS (klhr_amd/ess.py: FFT autocovariance, within/between-chain variance, Geyer initial-monotone "
                     "truncation), per chain-iteration, times the headline iterations/sec"}
    if not ok:
        out["reason"] = ("Geyer truncation lag >= draws / 10 for some coordinate and the two tau estimators of v disagree by more than 20 %: "
                         "the autocorrelation time is not resolved by this run")
    return out


def pooled_shard(rank, world, device, seed, dist):
    """N > 1 ranks: BASELINE config 5's per-GPU shard — funnel-100D, 131 072 chains per GPU — with pooled="full" adaptation:
    per window closure one ncclAllReduce (RCCL over xGMI) of 1 + D + D^2 doubles inside the library (klhr_comm_init /
    klhr_pool_allreduce).  Setup draws with the pooled closures, then 3 timed regions of 50 draws.  Raises when the
    communicator does not span all ranks (one device per rank), so that a silently skipped collective cannot pass."""
    res, D = run_workload("funnel", 100, 131072, 50, 0, 3, rank, world, device, seed + 29, dist, pooled="full")
    return res, D


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-baseline-worker":
        cpu_baseline_worker(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)    # untimed steps after the setup draws (the setup itself does not depend on this)
    ap.add_argument("--chains", type=int, default=65536, help="chains per GPU")
    ap.add_argument("--model", default="funnel", choices=["funnel", "ill-normal", "glmm-poisson", "normal"])
    ap.add_argument("--dim", type=int, default=10, help="model dimension (funnel: 1 + number of alphas)")
    ap.add_argument("--pooled", default=None, choices=["diag", "full"], help="pooled adaptation over all ranks during the setup draws (RCCL)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ess", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the short runs of the other BASELINE configurations")
    ap.add_argument("--seed", type=int, default=20261004)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print("bench.py: --gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)", file=sys.stderr)
            sys.exit(2)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    from klhr_amd import _lib
    ndev = _lib.device_count()
    if ndev < 1:
        raise SystemExit("bench.py: no HIP device visible (there is no CPU fallback)")
    device = local_rank % ndev      # one rank per GPU on the 8-GPU node; rehearsals on a 1-GPU box share device 0
    C = args.chains

    res, D = run_workload(args.model, args.dim, C, args.steps, args.warmup, REPEATS, rank, world, device, args.seed, dist, pooled=args.pooled)
    headline = args.model == "funnel" and D == 10
    out = {
        "metric": "MCMC iters/sec (all chains) on funnel-10D" if headline else f"MCMC iters/sec (all chains) on {args.model}-{D}D",
        "value": res["value"], "unit": "iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": res["ms_per_step"], "ms_per_step_min": res["ms_per_step_min"], "ms_per_step_max": res["ms_per_step_max"],
        "timed_regions": REPEATS, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic", "build_id": _lib.build_id(),
        "config": {"workload": f"{args.model}-{D}D KLHR-sinh (warmup=100, overrelaxed=False), {C} chains per GPU, draws recorded in HBM; "
                               f"timed after {SETUP_DRAWS} untimed setup draws (adaptation windows closed) + --warmup; "
                               f"median of {REPEATS} regions of --steps draws",
                   "chains_per_gpu": C, "chains_total": C * world, "dim": D,
                   "parallelism": f"chains sharded over {world} GPU(s), no collective in the timed region"
                                  + (f"; pooled={args.pooled} adaptation (RCCL all-reduce per window closure) during setup" if args.pooled else "")},
        "roofline": res["roofline"], "grad_evals_per_iter": res["grad_evals_per_iter"], "acceptance": res["acceptance"],
    }
    out["roofline"]["note"] = ("the path is fp64-VALU/latency bound at D = 10, not HBM bound (DESIGN.md section 6); fp64_valu.algorithmic counts "
                               "SURVEY 8d's flop-equivalents, fp64_valu.measured comes from the SQ counters of profiles/ (same build only)")
    if "pooled" in res:
        out["pooled"] = res["pooled"]
    rccl_failed = None
    if world > 1 and not args.no_configs:
        # every rank takes part (the collective needs them all); rank 0 reports.  A collective that never returns must not take the
        # headline with it: a watchdog prints the line measured so far (with rccl.error) and ends the process with a non-zero code.
        import threading

        def give_up():
            if rank == 0:
                out["rccl"] = {"error": f"the pooled shard (RCCL communicator / all-reduce over {world} ranks) did not finish within {POOLED_TIMEOUT_S} s"}
                print(json.dumps(out), flush=True)
                print("bench.py: the pooled shard's collective did not finish; see rccl.error in the line", file=sys.stderr, flush=True)
            os._exit(3)
        watchdog = threading.Timer(POOLED_TIMEOUT_S, give_up)
        watchdog.daemon = True
        watchdog.start()
        try:
            pr, pd = pooled_shard(rank, world, device, args.seed, dist)
            pc = pr["pooled"]
            rccl = {"ranks": pc["ranks"], "collective": pc["collective"], "doubles_per_allreduce": pc["doubles_per_allreduce"],
                    "closures_pooled": pc["closures_pooled"], "allreduce_ms": pc["allreduce_plus_fetch_ms"],
                    "comm_world": pc.get("comm_world"), "where": "untimed setup draws of the funnel-100D x 131072-per-GPU shard (BASELINE config 5)"}
            if pc.get("one_device_per_rank") and pc.get("comm_world") != world:
                rccl_failed = f"klhr_comm_world() = {pc.get('comm_world')} != world = {world}"
            out["rccl"] = rccl
            if rccl_failed:
                out["rccl"]["error"] = rccl_failed
            out.setdefault("configs", {})["funnel-100x131072-per-gpu-pooled-full"] = {
                "it_per_s": pr["value"], "ms_per_step": pr["ms_per_step"], "timed_regions": 3, "steps": 50, "kernel": pr["kernel"],
                "chains_total": 131072 * world, "pooled": "full", "roofline": {k: pr["roofline"][k] for k in ("achieved", "frac", "kernel", "launch_ms", "kernels", "whole_iteration")}}
        except Exception as e:          # reported in the line (rccl.error), on stderr AND through the exit code
            rccl_failed = f"{type(e).__name__}: {e}"
            out["rccl"] = {"error": rccl_failed}
        watchdog.cancel()
    if rank == 0:
        if not args.no_ess:
            if headline:
                out["ess_per_sec"] = ess_run(device, args.seed, res["value"])
            else:
                out["ess_per_sec"] = None
                out["ess_per_sec_reason"] = "estimated for the funnel-10D headline workload only"
        if world == 1 and not args.no_configs:
            # the other single-GPU BASELINE configurations, short runs after the same setup (3 regions of 50 draws each)
            cfgs = {}
            for key, (m, d, c) in {"ill-normal-100x16384": ("ill-normal", 100, 16384), "glmm-poisson-45x8192": ("glmm-poisson", 45, 8192),
                                   "funnel-100x131072": ("funnel", 100, 131072)}.items():
                r, dd = run_workload(m, d, c, 50, 0, 3, 0, 1, device, args.seed + 17, None)
                cfgs[key] = {"it_per_s": r["value"], "ms_per_step": r["ms_per_step"], "ms_per_step_min": r["ms_per_step_min"],
                             "ms_per_step_max": r["ms_per_step_max"], "timed_regions": 3, "steps": 50, "kernel": r["kernel"],
                             "roofline": {k: r["roofline"][k] for k in ("achieved", "frac", "traffic", "kernel", "longest_kernel", "launch_ms", "iters_per_launch",
                                                                       "algorithmic_bytes_per_chain_iter", "kernels", "kernel_times_from", "whole_iteration",
                                                                       "traffic_stale", "traffic_reason", "traffic_over_algorithmic") if k in r["roofline"]},
                             "fp64_valu": r["roofline"]["fp64_valu"], "grad_evals_per_iter": r["grad_evals_per_iter"],
                             "acceptance": r["acceptance"]}
            out["configs"] = cfgs
        if world == 1 and not args.no_cpu_baseline:
            ncpu = os.cpu_count() or 1
            try:
                ncpu = len(os.sched_getaffinity(0))          # the cores this process may actually use ...
            except (AttributeError, OSError):
                pass
            quota = None
            try:                                             # ... and the CPU time its cgroup grants (a 1-GPU box hands out a share of the host)
                q, per = open("/sys/fs/cgroup/cpu.max").read().split()
                if q != "max":
                    quota = max(1, int(float(q) / float(per)))
            except (OSError, ValueError):
                pass
            if quota:
                ncpu = min(ncpu, quota)
            threads = min(ncpu, 16)
            n_cpu_chains, cpu_iters = 256 * threads, 200        # ~1.2 s wall on 16 threads = ~19 s of CPU work
            v = cpu_baseline(n_cpu_chains, 100, cpu_iters, threads)
            v1 = cpu_baseline(64, 100, 200, 1)                   # ~0.4 s: one thread, 64 chains one after the other
            allc = None
            if ncpu > threads:                                   # "all-core total" — every core this process is allowed
                va = cpu_baseline(64 * ncpu, 100, 100, ncpu)
                allc = {"value": va, "unit": "iterations/s", "cores": ncpu,
                        "sample": f"{64 * ncpu} funnel-10D chains x 100 draws after 100 warmup draws, {ncpu} threads (all cores of this process's affinity mask)"}
            out["cpu_baseline"] = {"value": v, "unit": "iterations/s", "cores": threads, "kind": "port",
                                   "sample": f"CPU oracle (C restatement of the reference path, OpenMP over chains): {n_cpu_chains} funnel-10D chains x {cpu_iters} draws after 100 warmup draws",
                                   "single_thread": {"value": v1, "unit": "iterations/s", "sample": "64 funnel-10D chains x 200 draws after 100 warmup draws, 1 thread"},
                                   "all_cores": allc, "host_cpus_usable": ncpu, "cgroup_cpu_quota": quota,
                                   "all_cores_note": None if allc else f"this process may use {ncpu} CPUs (affinity mask and cgroup quota): the {threads}-thread figure is the all-core total",
                                   "cpu_model": cpu_model_string(), "host_cpus": os.cpu_count(),
                                   "reference_python": {"it_per_s_per_chain": [59.4, 78.4], "unit": "iterations/s per chain",
                                                        "where": "BASELINE.md section 2: the reference's own klhr_sinh.py, NumPy stand-in model, 1 core of the build container (the reference cannot travel to the GPU box)"}}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rccl_failed:
        print(f"bench.py: the pooled shard's collective did not run over all ranks: {rccl_failed}", file=sys.stderr)
        sys.exit(3)


if __name__ == "__main__":
    main()

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
ESS/sec comes from a separate, fixed post-warm-up run (2 048 chains x 1 000 draws, outside the timed regions): ESS per
chain-iteration x the measured iterations/sec.  Rank 0 prints ONE JSON line.
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
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        return json.load(open(tpath)).get(key)
    except Exception:
        return None


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
        comm = pooled_setup_draws(s, SETUP_DRAWS, rank, world, dist, mode=pooled)
    else:
        s.step(SETUP_DRAWS)                 # the sampler's own adaptation: untimed setup, whatever --warmup says
    if warm > 0:
        s.step(warm)
    s.reserve_draws(steps)                  # the output buffer (np.zeros((M, D)) in experiment_funnel.py:48) exists before the clock starts
    region = []
    for _ in range(repeats):
        s.sync()
        barrier()
        s.sync()
        t0 = time.perf_counter()
        s.step(steps, record_draws=True, sync=False)
        s.sync()
        barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            import torch
            t = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t[0])
        region.append(dt)
    dt_med = float(np.median(region))
    total_iters = world * C * steps
    value = total_iters / dt_med
    # ---- per-kernel durations from HIP events on the launch stream (untimed pass; split pipeline: one chain group)
    s.kernel_profiling(True)
    # (monolithic kernels run the whole call in one or a few launches: profile a call of the timed length, so that every
    #  dispatch a rocprofv3 pass sees at the end of the run covers the same number of iterations)
    n_prof = min(steps, profile_iters) if len(s.kernel_names()) > 1 else steps
    s.step(n_prof, record_draws=True)
    kms, kn = s.kernel_times()
    s.kernel_profiling(False)
    names = s.kernel_names()
    b_iter = algorithmic_bytes_per_iter(D)
    ge = s.get("grad_evals").astype(np.float64)
    cnt = s.get("counters")[:, 0].astype(np.float64)
    grad_evals = float((ge / np.maximum(cnt, 1)).mean())
    kernels = {}
    for nm, ms, n in zip(names, kms, kn):
        if n:
            kernels[nm] = {"mean_ms": ms, "launches": n}
    # the roofline's kernel: the one that moves the iteration's algorithmic bytes — the hot kernel of the monolithic mappings, the
    # streaming kernel of the split pipeline (its fit kernels work on a few doubles per chain; when they take as long as the
    # streaming kernel, which of the two is "dominant" by time flips from run to run: `longest` says which one it was)
    longest = max(kernels, key=lambda k: kernels[k]["mean_ms"] * kernels[k]["launches"])
    dom = names[0] if names[0] in kernels else longest
    # iterations covered by one launch of the dominant kernel: the monolithic kernels run several per launch
    iters_per_launch = (n_prof / kernels[dom]["launches"]) if len(names) == 1 else 1.0
    launch_s = kernels[dom]["mean_ms"] / 1e3
    achieved_gbs = b_iter * C * iters_per_launch / launch_s / 1e9
    tkey = f"{model}-{D}-{C}"
    te = traffic_entry(tkey)
    traffic = te["hbm_bytes_per_launch"] * (iters_per_launch / te["iters_per_launch"]) if te else None
    fe = flop_equiv_per_iter(D, grad_evals)
    out = {
        "value": value, "ms_per_step": dt_med * 1e3 / steps,
        "ms_per_step_min": min(region) * 1e3 / steps, "ms_per_step_max": max(region) * 1e3 / steps, "timed_regions": repeats,
        "roofline": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved_gbs / HBM_PEAK_GBS,
                     "traffic": traffic, "kernel": dom, "longest_kernel": longest, "algorithmic_bytes_per_chain_iter": b_iter,
                     "launch_ms": launch_s * 1e3, "iters_per_launch": iters_per_launch, "kernels": kernels,
                     "whole_iteration": {"achieved": b_iter * value / world / 1e9, "frac": b_iter * value / world / 1e9 / HBM_PEAK_GBS,
                                         "note": "algorithmic bytes x measured iterations/s (all kernels of an iteration, per GPU)"},
                     "fp64_valu": {"achieved_tflops": value / world * fe / 1e12, "peak_tflops": FP64_VALU_PEAK_TFLOPS,
                                   "frac": value / world * fe / 1e12 / FP64_VALU_PEAK_TFLOPS, "flop_equivalents_per_chain_iter": fe}},
        "grad_evals_per_iter": grad_evals, "acceptance": float(s.get("acc").mean()), "kernel": s.kernel_name(),
    }
    if traffic is not None:
        out["roofline"]["traffic_source"] = te.get("source")
    if comm is not None:
        out["pooled"] = comm
    s.close()
    return out, D


def ess_run(device, seed, value):
    """ESS of the funnel-10D chains from a fixed post-warm-up run, outside every timed region: 2 048 chains, the same setup
    draws, then 1 000 recorded draws.  ESS/sec = ESS per chain-iteration x the headline iterations/sec (all chains)."""
    from klhr_amd import _lib
    from klhr_amd.ess import ess
    nch, ndraw = 2048, 1000
    data, D = model_data("funnel", 10)
    s = _lib.Sampler(_lib.MODEL_IDS["funnel"], data, nch, device=device, warmup=100)
    s.seed(np.uint64(seed) + np.uint64(7_000_000) + np.arange(nch, dtype=np.uint64))
    s.step(SETUP_DRAWS)
    s.sync()
    t0 = time.perf_counter()
    s.step(ndraw, record_draws=True)
    dt = time.perf_counter() - t0
    dr = s.draws(0, ndraw)
    s.close()
    e_v = ess(dr[:, :, 0])
    e_min = min(ess(dr[:, :, j]) for j in range(D))
    per_iter_v, per_iter_min = e_v / (nch * ndraw), e_min / (nch * ndraw)
    return {"v": per_iter_v * value, "min_coord": per_iter_min * value,
            "ess_per_chain_iteration": {"v": per_iter_v, "min_coord": per_iter_min},
            "estimated_on": {"chains": nch, "draws": ndraw, "after_setup_draws": SETUP_DRAWS, "its_own_it_per_s": nch * ndraw / dt},
            "method": "rank-normalisation-free multi-chain ESS (klhr_amd/ess.py: FFT autocovariance, Geyer initial-monotone "
                      "truncation, combined across chains), times the headline iterations/sec"}


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
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{args.model}-{D}D KLHR-sinh (warmup=100, overrelaxed=False), {C} chains per GPU, draws recorded in HBM; "
                               f"timed after {SETUP_DRAWS} untimed setup draws (adaptation windows closed) + --warmup; "
                               f"median of {REPEATS} regions of --steps draws",
                   "chains_per_gpu": C, "chains_total": C * world, "dim": D,
                   "parallelism": f"chains sharded over {world} GPU(s), no collective in the timed region"
                                  + (f"; pooled={args.pooled} adaptation (RCCL all-reduce per window closure) during setup" if args.pooled else "")},
        "roofline": res["roofline"], "grad_evals_per_iter": res["grad_evals_per_iter"], "acceptance": res["acceptance"],
    }
    out["roofline"]["note"] = ("the path is fp64-VALU/latency bound at D = 10, not HBM bound (DESIGN.md section 6); fp64_valu counts "
                               "algorithmic flop-equivalents (SURVEY section 8d)")
    if "pooled" in res:
        out["pooled"] = res["pooled"]
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
                             "roofline": {k: r["roofline"][k] for k in ("achieved", "frac", "traffic", "kernel", "launch_ms", "iters_per_launch",
                                                                       "algorithmic_bytes_per_chain_iter", "kernels", "whole_iteration")},
                             "fp64_valu_frac": r["roofline"]["fp64_valu"]["frac"], "grad_evals_per_iter": r["grad_evals_per_iter"],
                             "acceptance": r["acceptance"]}
            out["configs"] = cfgs
        if world == 1 and not args.no_cpu_baseline:
            threads = min(os.cpu_count() or 1, 16)
            n_cpu_chains, cpu_iters = 256 * threads, 200        # ~1.2 s wall on 16 threads = ~19 s of CPU work
            v = cpu_baseline(n_cpu_chains, 100, cpu_iters, threads)
            v1 = cpu_baseline(64, 100, 200, 1)                   # ~0.4 s: one thread, 64 chains one after the other
            out["cpu_baseline"] = {"value": v, "unit": "iterations/s", "cores": threads, "kind": "port",
                                   "sample": f"CPU oracle (C restatement of the reference path, OpenMP over chains): {n_cpu_chains} funnel-10D chains x {cpu_iters} draws after 100 warmup draws",
                                   "single_thread": {"value": v1, "unit": "iterations/s", "sample": "64 funnel-10D chains x 200 draws after 100 warmup draws, 1 thread"},
                                   "cpu_model": cpu_model_string(), "host_cpus": os.cpu_count(),
                                   "reference_python": {"it_per_s_per_chain": [59.4, 78.4], "unit": "iterations/s per chain",
                                                        "where": "BASELINE.md section 2: the reference's own klhr_sinh.py, NumPy stand-in model, 1 core of the build container (the reference cannot travel to the GPU box)"}}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

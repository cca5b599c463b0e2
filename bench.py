#!/usr/bin/env python3
"""bench.py — headline benchmark: MCMC iterations/sec (all chains) + ESS/sec on the funnel-10D target.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

A "step" is one KLHRSINH.draw() for every chain of the workload (BASELINE.json configs[1]: funnel 10-dim, 65 536
chains per MI355X, log density as a HIP device function).  Chains shard across GPUs with no data-path collective
(weak scaling: 65 536 chains per GPU); torch.distributed (gloo, CPU tensors) is used only for the barrier and the
max-over-ranks timing the contract asks for.  Inputs (chain state) are resident in HBM when the timed region starts;
draws are recorded into an HBM buffer inside the timed region and never copied to the host there.
Rank 0 prints ONE JSON line.
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


def algorithmic_bytes_per_iter(D, J=2):
    """reads (5+2J)D+J doubles, writes (4+J)D doubles (draw recorded), + RNG state and scalars."""
    return 8 * ((5 + 2 * J) * D + J + (4 + J) * D) + 112


def flop_equiv_per_iter(D):
    """~14 KL evals x 8 nodes + ~8.5 single evals of (9D+10 flops + 1 transcendental) + ~40 flops and 3
    transcendentals per node + ~11 BFGS updates of ~250 flops; a transcendental counted as 25 flops."""
    node_evals = 14.0 * 8 + 8.5
    return node_evals * (9 * D + 10 + 25) + 14.0 * 8 * (40 + 3 * 25) + 11 * 250


def cpu_baseline(chains, warm, iters, threads):
    """The CPU oracle (oracle/klhr_oracle.c, a validated restatement of the reference path) timed on host cores."""
    os.environ["OMP_NUM_THREADS"] = str(threads)
    from oracle import oracle as O
    O.build()
    cs = [O.Chain("funnel", {"D": 9}, seed=10_000 + i, warmup=100) for i in range(chains)]
    O.draw_batch(cs, warm)
    t0 = time.perf_counter()
    O.draw_batch(cs, iters)
    dt = time.perf_counter() - t0
    return chains * iters / dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=200)   # one launch of the same length as the timed one (closes both adaptation windows)
    ap.add_argument("--chains", type=int, default=65536, help="chains per GPU")
    ap.add_argument("--model", default="funnel", choices=["funnel", "ill-normal", "glmm-poisson", "normal"])
    ap.add_argument("--dim", type=int, default=10, help="model dimension (funnel: 1 + number of alphas)")
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
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    from klhr_amd import _lib
    import json as _json
    if args.model == "funnel":
        data, D = np.array([float(args.dim - 1)]), args.dim
    elif args.model == "glmm-poisson":
        d = _json.load(open(os.path.join(ROOT, "klhr_amd", "stan", "glmm-poisson.json")))
        data, D = np.concatenate([[float(d["n"])], d["year"], d["C"]]), d["n"] + 5
    else:
        data, D = np.array([float(args.dim)]), args.dim
    C = args.chains
    ndev = _lib.device_count()
    if ndev < 1:
        raise SystemExit("bench.py: no HIP device visible (there is no CPU fallback)")
    device = local_rank % ndev      # one rank per GPU on the 8-GPU node; rehearsals on a 1-GPU box share device 0
    s = _lib.Sampler(_lib.MODEL_IDS[args.model], data, C, device=device, warmup=100)   # experiment_funnel.py:21 default
    seeds = np.uint64(args.seed) + np.uint64(rank) * np.uint64(C) + np.arange(C, dtype=np.uint64)
    s.seed(seeds)

    def barrier():
        if dist is not None:
            dist.barrier()

    # ---- warmup (untimed): W iterations, which also closes the adaptation windows when W >= 100
    if args.warmup > 0:
        s.step(args.warmup)
    s.reserve_draws(args.steps)     # the output buffer (np.zeros((M, D)) in experiment_funnel.py:48) exists before the clock starts
    s.sync()
    barrier()
    s.sync()
    t0 = time.perf_counter()
    s.step(args.steps, record_draws=True, sync=False)
    s.sync()
    barrier()
    t1 = time.perf_counter()
    dt = t1 - t0
    kernel_ms, n_launch = s.timing()
    if dist is not None:
        import torch
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])
    total_iters = world * C * args.steps
    value = total_iters / dt

    # ---- roofline of the dominant kernel (klhr_iter_kernel), from HIP events recorded on the launch stream
    b_iter = algorithmic_bytes_per_iter(D)
    launch_s = kernel_ms / 1e3 / max(n_launch, 1)
    iters_per_launch = args.steps / max(n_launch, 1)
    achieved_gbs = b_iter * C * iters_per_launch / launch_s / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            key = f"{args.model}-{D}-{C}"
            if key in tj:
                traffic = tj[key]["hbm_bytes_per_launch"] * (iters_per_launch / tj[key]["iters_per_launch"])
        except Exception:
            traffic = None
    out = {
        "metric": "MCMC iters/sec (all chains) on funnel-10D" if (args.model == "funnel" and D == 10) else f"MCMC iters/sec (all chains) on {args.model}-{D}D",
        "value": value, "unit": "iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{args.model}-{D}D KLHR-sinh (warmup=100, overrelaxed=False), {C} chains per GPU, draws recorded in HBM",
                   "chains_per_gpu": C, "chains_total": C * world, "dim": D, "parallelism": f"chains sharded over {world} GPU(s), no collective"},
        "roofline": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic,
                     "kernel": s.kernel_name(), "algorithmic_bytes_per_chain_iter": b_iter,
                     "launch_ms": launch_s * 1e3, "launches": n_launch,
                     "fp64_valu": {"achieved_tflops": value / world * flop_equiv_per_iter(D) / 1e12, "peak_tflops": FP64_VALU_PEAK_TFLOPS,
                                   "frac": value / world * flop_equiv_per_iter(D) / 1e12 / FP64_VALU_PEAK_TFLOPS,
                                   "flop_equivalents_per_chain_iter": flop_equiv_per_iter(D)},
                     "note": "the path is fp64-VALU/latency bound, not HBM bound (DESIGN.md section 6); fp64_valu counts algorithmic flop-equivalents (SURVEY section 8d)"},
    }
    if rank == 0:
        # grad-evals per iteration and acceptance (cheap state reads, outside the timed region)
        ge = s.get("grad_evals").astype(np.float64)
        cnt = s.get("counters")[:, 0].astype(np.float64)
        out["grad_evals_per_iter"] = float((ge / np.maximum(cnt, 1)).mean())
        out["acceptance"] = float(s.get("acc").mean())
        if not args.no_ess:
            from klhr_amd.ess import ess
            nsub = min(C, 2048)
            dr = s.draws(0, args.steps)[:, :nsub, :]
            e_v = ess(dr[:, :, 0])
            e_min = min(ess(dr[:, :, j]) for j in range(D))
            out["ess_per_sec"] = {"v": e_v * (C * world / nsub) / dt, "min_coord": e_min * (C * world / nsub) / dt,
                                  "estimated_on_chains": nsub}
        if world == 1 and not args.no_cpu_baseline:
            threads = min(os.cpu_count() or 1, 16)
            n_cpu_chains, cpu_iters = 256 * threads, 200        # ~1.2 s wall on 16 threads = ~19 s of CPU work
            v = cpu_baseline(n_cpu_chains, 100, cpu_iters, threads)
            out["cpu_baseline"] = {"value": v, "unit": "iterations/s", "cores": threads, "kind": "port",
                                   "sample": f"CPU oracle (C restatement of the reference path, OpenMP over chains): {n_cpu_chains} funnel-10D chains x {cpu_iters} draws after 100 warmup draws"}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

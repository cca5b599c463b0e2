"""Counterpart of the reference's experiment_relaxationtime.py (:16-106) on stan/earnings (4 parameters, 1192 data
points): trace plots of (beta_0, beta_1, sigma, s) from the 0.1 N(0, 1) start, i.e. how many draws the sampler needs to
reach the typical set; prints acceptance rate, MSJD, post-warm-up means/sds and the gradient count.

    python -m klhr_amd.experiment_relaxationtime -M 2000 -w 1000 -v klhr_sinh
"""
import sys
from pathlib import Path

import click
import numpy as np

from .bsmodel import BSModel
from .klhr import KLHR
from .klhr_sinh import KLHRSINH
from .slice import Slice
from .sub_klhr_sinh import SUBKLHRSINH

STAN = Path(__file__).resolve().parent / "stan"
SAMPLERS = {"klhr_sinh": KLHRSINH, "klhr": KLHR, "sub_klhr_sinh": SUBKLHRSINH, "slice": Slice}


@click.command()
@click.option("-M", "--iterations", "M", type=int, default=1_000)
@click.option("-w", "--warmup", "warmup", type=int, default=100)
@click.option("--windowsize", "windowsize", type=int, default=25)
@click.option("--windowscale", "windowscale", type=int, default=2)
@click.option("-l", "--amnesia", "l", type=int, default=2)
@click.option("-J", "J", type=int, default=2)
@click.option("-r", "--replication", "rep", type=int, default=0)
@click.option("-v", "--verbose", "verbose", is_flag=True)
@click.option("-s", "--scale_dir_cov", "scale_dir_cov", is_flag=True)
@click.option("-o", "--overrelaxed", "overrelaxed", is_flag=True)
@click.option("-e1", "--eigen_method_one", "eigen_method_one", is_flag=True)
@click.option("--seed", type=int, default=None)
@click.option("--outdir", type=click.Path(), default="experiments/relaxationtime")
@click.option("--no-plots", is_flag=True)
@click.argument("algorithm", type=str)
def main(M, warmup, windowsize, windowscale, l, J, rep, verbose, scale_dir_cov, overrelaxed, eigen_method_one, seed, outdir,
         no_plots, algorithm):
    if algorithm not in SAMPLERS:
        print(f"Unknown algorithm {algorithm}")
        print("Available algorithms: klhr, klhr_sinh, sub_klhr_sinh, or slice")
        sys.exit(0)
    bs_model = BSModel(stan_file=STAN / "earnings.stan", data_file=STAN / "earnings.json")
    algo = SAMPLERS[algorithm](bs_model, warmup=warmup, windowsize=windowsize, windowscale=windowscale, J=J, l=l,
                               scale_dir_cov=scale_dir_cov, overrelaxed=overrelaxed, eigen_method_one=eigen_method_one, seed=seed)
    draws = algo.sample(M)
    if verbose:
        print(f"Acceptance rate: {np.round(algo.acceptance_probability, 2)}")
        msjd = np.mean(np.linalg.norm(np.diff(draws, axis=0), axis=1))
        print(f"MSJD: {np.round(msjd, 2)}")
        print(np.mean(draws[warmup:, :], axis=0))
        print(np.std(draws[warmup:, :], axis=0))
        print(f"Number gradients: {algo.grad_evals}")
    if no_plots:
        return
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    out = Path(outdir)
    out.mkdir(parents=True, exist_ok=True)
    idx = np.arange(M)
    fig, axs = plt.subplots(2, 2, figsize=(14, 6))
    for ax, k, lab in zip(axs.ravel(), range(4), (r"$\beta_0$", r"$\beta_1$", r"$\sigma$", r"$s$")):
        ax.plot(idx, draws[:, k])
        ax.set_ylabel(lab)
    plt.tight_layout()
    plt.savefig(out / f"{algorithm}_{windowsize}_{windowscale}_{l}_{J}_{scale_dir_cov}_{overrelaxed}_{eigen_method_one}_{rep:0>2}.png")
    plt.close()


if __name__ == "__main__":
    main()

"""Counterpart of the reference's experiment_ar1.py (:16-117) on stan/ar1 (100-dimensional AR(1) Gaussian, unit
marginal variances): per-coordinate histograms against N(0, 1), RMSE of the means and variances.  Same flags and
defaults (windowsize 25, amnesia l = 2); extra --chains / --seed.  (The reference's verbose branch reads a
non-existent `minimization_failure_rate` attribute, experiment_ar1.py:91 — not reproduced.)

    python -m klhr_amd.experiment_ar1 -M 1000 -w 100 -v klhr_sinh
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
@click.option("-M", "--iterations", "M", type=int, default=1_000, help="number of iterations")
@click.option("-w", "--warmup", "warmup", type=int, default=100, help="number of warmup iterations")
@click.option("--windowsize", "windowsize", type=int, default=25, help="set window size")
@click.option("--windowscale", "windowscale", type=int, default=2, help="set window scale")
@click.option("-l", "--amnesia", "l", type=int, default=2, help="set the amnesia parameter for OnlinePCA")
@click.option("-J", "J", type=int, default=2, help="number of eigenvectors")
@click.option("-r", "--replication", "rep", type=int, default=0, help="replication number for naming output files")
@click.option("-v", "--verbose", "verbose", is_flag=True, help="print information during run")
@click.option("-s", "--scale_dir_cov", "scale_dir_cov", is_flag=True)
@click.option("-o", "--overrelaxed", "overrelaxed", is_flag=True)
@click.option("-e1", "--eigen_method_one", "eigen_method_one", is_flag=True)
@click.option("--chains", type=int, default=4)
@click.option("--seed", type=int, default=None)
@click.option("--outdir", type=click.Path(), default="experiments/ar1")
@click.option("--no-plots", is_flag=True)
@click.argument("algorithm", type=str)
def main(M, warmup, windowsize, windowscale, l, J, rep, verbose, scale_dir_cov, overrelaxed, eigen_method_one, chains, seed,
         outdir, no_plots, algorithm):
    if algorithm not in SAMPLERS:
        print(f"Unknown algorithm {algorithm}")
        print("Available algorithms: klhr, klhr_sinh, sub_klhr_sinh, or slice")
        sys.exit(0)
    bs_model = BSModel(stan_file=STAN / "ar1.stan", data_file=STAN / "ar1.json")
    algo = SAMPLERS[algorithm](bs_model, warmup=warmup, windowsize=windowsize, windowscale=windowscale, l=l, J=J,
                               scale_dir_cov=scale_dir_cov, overrelaxed=overrelaxed, eigen_method_one=eigen_method_one,
                               chains=chains, seed=seed)
    thetas = algo.draws(M)[warmup:].reshape(-1, algo.D)
    v = np.var(thetas, ddof=1, axis=0)
    m = np.mean(thetas, axis=0)
    rmse_var, rmse_mean = np.sqrt(np.mean((v - 1) ** 2)), np.sqrt(np.mean(m ** 2))
    print(f"RMSE(mean) = {rmse_mean:.4f}, RMSE(var) = {rmse_var:.4f}")
    if verbose:
        print(f"Acceptance rate: {algo.acceptance_probability}")
        print(f"maximum absolute mean: {np.max(np.abs(m)):.4f}")
        print(f"minimum variance: {np.min(v):.4f}")
    if no_plots:
        return
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    import scipy.stats as st
    out = Path(outdir)
    out.mkdir(parents=True, exist_ok=True)
    plt.clf()
    for d in range(algo.D):
        plt.hist(thetas[:, d], histtype="step", density=True, color="#0072B2", alpha=0.1)
    x = np.linspace(-4, 4, 301)
    plt.plot(x, st.norm().pdf(x), linestyle="dashed", color="#D55E00")
    plt.title(f"RMSE(mean) = {rmse_mean:.4f}, RMSE(var) = {rmse_var:.4f}")
    plt.tight_layout()
    plt.savefig(out / f"{algorithm}_{windowsize}_{windowscale}_{l}_{J}_{scale_dir_cov}_{overrelaxed}_{eigen_method_one}_{rep:0>2}.png")
    plt.close()


if __name__ == "__main__":
    main()

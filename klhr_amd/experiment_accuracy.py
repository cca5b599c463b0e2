"""Counterpart of the reference's experiment_accuracy.py (:24-165) on stan/normal: running RMSE of mean and variance
for KLHR-sinh vs MH(0.09) vs iid draws, and log-density histograms.  Same flags/defaults; sampler + comparator on GPU.

    python -m klhr_amd.experiment_accuracy -M 1000 klhr_sinh
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
from .mh import MH
from .onlinemoments import OnlineMoments

STAN = Path(__file__).resolve().parent / "stan"


def running_rmse(draws):
    """Per-m RMSE of the running mean (target 0) and running variance (target 1): experiment_accuracy.py:93-98."""
    M, D = draws.shape
    om = OnlineMoments(D)
    rm, rv = np.zeros(M), np.zeros(M)
    for m in range(M):
        om.update(draws[m])
        rm[m] = np.sqrt(np.mean(om.mean() ** 2))
        rv[m] = np.sqrt(np.mean((om.var() - 1) ** 2))
    return rm, rv


@click.command()
@click.option("-M", "--iterations", "M", type=int, default=1_000, help="number of iterations")
@click.option("-w", "--warmup", "warmup", type=int, default=100, help="set value from which RMSEs are plot")
@click.option("-v", "--verbose", "verbose", is_flag=True, help="print information during run")
@click.option("-s", "--scale_dir_cov", "scale_dir_cov", is_flag=True)
@click.option("-o", "--overrelaxed", "overrelaxed", is_flag=True)
@click.option("-e1", "--eigen_method_one", "eigen_method_one", is_flag=True)
@click.option("--seed", type=int, default=None)
@click.option("--outdir", type=click.Path(), default="experiments/accuracy")
@click.option("--no-plots", is_flag=True)
@click.argument("algorithm", type=str)
def main(M, warmup, verbose, scale_dir_cov, overrelaxed, eigen_method_one, seed, outdir, no_plots, algorithm):
    if algorithm not in ("klhr_sinh", "klhr", "sub_klhr_sinh", "slice"):
        print(f"Unknown algorithm {algorithm}")
        print("Available algorithms: klhr, klhr_sinh, sub_klhr_sinh, or slice")
        sys.exit(0)
    Sampler = {"klhr_sinh": KLHRSINH, "klhr": KLHR, "sub_klhr_sinh": SUBKLHRSINH, "slice": Slice}[algorithm]
    bs_model = BSModel(stan_file=STAN / "normal.stan", data_file=STAN / "normal.json")
    algo = Sampler(bs_model, warmup=warmup, scale_dir_cov=scale_dir_cov, overrelaxed=overrelaxed,
                    eigen_method_one=eigen_method_one, seed=seed)
    mh = MH(bs_model, 0.09, seed=None if seed is None else seed + 1)
    draws = algo.sample(M)                       # row 0 = initial point (mcmc.py:34)
    mh_draws = mh.draws(M)
    rng = np.random.default_rng(seed)
    iid = rng.multivariate_normal(np.zeros(algo.D), np.eye(algo.D), size=M)
    rm_k, rv_k = running_rmse(draws)
    rm_m, rv_m = running_rmse(mh_draws)
    ld_k, ld_m, ld_i = (bs_model.log_density(x) for x in (draws, mh_draws, iid))
    if verbose:
        print(f"Acceptance rate: {algo.acceptance_probability}")
        print(f"MSJD: {np.round(np.mean(np.linalg.norm(np.diff(draws, axis=0), axis=1)), 2)}")
        print(f"means: {draws.mean(axis=0)}")
        print(f"vars: {draws.var(axis=0, ddof=1)}")
    print(f"final RMSE  klhr_sinh: mean {rm_k[-1]:.4f} var {rv_k[-1]:.4f} | MH: mean {rm_m[-1]:.4f} var {rv_m[-1]:.4f}")
    if no_plots:
        return
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    out = Path(outdir)
    out.mkdir(parents=True, exist_ok=True)
    tag = f"{algorithm}_{warmup}_{scale_dir_cov}_{overrelaxed}_{eigen_method_one}"
    origin = min(10 ** 2, M // 2)
    mdx = np.arange(M)
    plt.clf()
    plt.plot(mdx[origin:], rm_k[origin:], label="KLHR: mean", linestyle="dotted", color="#0072B2", linewidth=2)
    plt.plot(mdx[origin:], rv_k[origin:], label="KLHR: var", linestyle=(0, (1, 5)), color="#D55E00", linewidth=2)
    plt.plot(mdx[origin:], rm_m[origin:], label="MH: mean", linestyle="dashed", color="#009E73", linewidth=2)
    plt.plot(mdx[origin:], rv_m[origin:], label="MH: var", linestyle=(0, (5, 5)), color="#F0E442", linewidth=2)
    plt.plot([origin, M], [1 / np.sqrt(origin), 1 / np.sqrt(M)], linestyle="solid", color="black", alpha=0.2)
    plt.yscale("log"); plt.xscale("log"); plt.ylabel("RMSE"); plt.xlabel("iteration"); plt.legend(); plt.tight_layout()
    plt.savefig(out / f"{tag}_rmse.png")
    plt.clf()
    for ld, lab in ((ld_k, "KLHR"), (ld_m, "MH"), (ld_i, "iid")):
        plt.hist(ld[origin:], histtype="step", density=True, label=lab)
    plt.xlabel("log_density"); plt.legend(); plt.tight_layout()
    plt.savefig(out / f"{tag}_histogram_log_density.png")
    plt.close()


if __name__ == "__main__":
    main()

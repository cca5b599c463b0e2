"""Counterpart of the reference's experiment_funnel.py (:19-73): same CLI flags and defaults, same outputs (scatter of
alpha_1 vs v and histogram of v against N(0, 3)), with the sampler running on the GPU and an extra --chains option.

    python -m klhr_amd.experiment_funnel -M 1000 -w 100 klhr_sinh
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


@click.command()
@click.option("-M", "--iterations", "M", type=int, default=1_000, help="number of iterations")
@click.option("-w", "--warmup", "warmup", type=int, default=100, help="number of warmup iterations")
@click.option("-v", "--verbose", "verbose", is_flag=True, help="print information during run")
@click.option("-s", "--scale_dir_cov", "scale_dir_cov", is_flag=True, help="scale covariance matrix used to select a random direction")
@click.option("-o", "--overrelaxed", "overrelaxed", is_flag=True, help="use overrelaxed proposals in metropolis step")
@click.option("-e1", "--eigen_method_one", "eigen_method_one", is_flag=True, help="Use option one for utilizing eigenvectors to select a direction")
@click.option("--chains", type=int, default=4, help="independent chains run on the GPU (extension)")
@click.option("--seed", type=int, default=None, help="base seed (chain c uses seed + c)")
@click.option("--outdir", type=click.Path(), default="experiments/funnel")
@click.option("--no-plots", is_flag=True)
@click.argument("algorithm", type=str)
def main(M, warmup, verbose, scale_dir_cov, overrelaxed, eigen_method_one, chains, seed, outdir, no_plots, algorithm):
    if algorithm not in ("klhr_sinh", "klhr", "sub_klhr_sinh", "slice"):
        print(f"Unknown algorithm {algorithm}")
        print("Available algorithms: klhr, klhr_sinh, sub_klhr_sinh, or slice")
        sys.exit(0)
    Sampler = {"klhr_sinh": KLHRSINH, "klhr": KLHR, "sub_klhr_sinh": SUBKLHRSINH, "slice": Slice}[algorithm]
    bs_model = BSModel(stan_file=STAN / "funnel.stan", data_file=STAN / "funnel.json")
    algo = Sampler(bs_model, warmup=warmup, scale_dir_cov=scale_dir_cov, overrelaxed=overrelaxed,
                    eigen_method_one=eigen_method_one, chains=chains, seed=seed)
    thetas = algo.draws(M)                      # (M, chains, D): M draws, no initial point (experiment_funnel.py:49-52)
    if verbose:
        print(f"Acceptance rate: {algo.acceptance_probability}")
    post = thetas[warmup:].reshape(-1, algo.D)
    print(f"v = theta[0]: mean {post[:, 0].mean():+.3f} sd {post[:, 0].std():.3f} (target N(0, 3)); draws {post.shape[0]}")
    if no_plots:
        return
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    import scipy.stats as st
    out = Path(outdir)
    out.mkdir(parents=True, exist_ok=True)
    tag = f"{scale_dir_cov}_{overrelaxed}_{eigen_method_one}_{algorithm}"
    plt.clf()
    plt.scatter(post[:, 1], post[:, 0], color="#0072B2", alpha=0.1)
    plt.tight_layout()
    plt.savefig(out / f"scatter_{tag}.png")
    plt.clf()
    plt.hist(post[:, 0], histtype="step", density=True, linewidth=2)
    grid = np.linspace(-10, 10, 101)
    plt.plot(grid, st.norm(loc=0, scale=3).pdf(grid), color="#D55E00")
    plt.tight_layout()
    plt.savefig(out / f"histogram_{tag}.png")
    plt.close()


if __name__ == "__main__":
    main()

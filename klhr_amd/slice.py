"""`Slice` — the reference's slice sampler along the KLHR direction draw (slice.py:12-183): Neal's univariate slice
sampler (stepping out, shrinkage) on g(x) = logp(theta + x rho), rho from the same windowed OnlineMoments / CCIPCA
adaptation as KLHRSINH.  Same kernels (`klhr_config.variant = 3`): direction draw, window closures and streaming updates
are shared; the two BFGS fits and the MH step are replaced by the slice step.

Constructor keywords and defaults are the reference's (slice.py:14-32).  Only `m = inf` (unlimited stepping out) is
available: the reference's finite-`m` branch uses undefined names and raises NameError.  `overrelaxed` is accepted and
ignored, as in the reference.  `grad_evals` counts log-density evaluations (a TODO in the reference)."""
import numpy as np

from . import _lib
from .klhr_sinh import KLHRSINH
from .mcmc import MCMCBase
from .windowedadaptation import WindowedAdaptation


class Slice(KLHRSINH):
    def __init__(self, bsmodel, theta=None, seed=None, w=1, m=np.inf, lower=-np.inf, upper=np.inf, J=2, l=4, initscale=0.1,
                 warmup=1_000, windowsize=50, windowscale=2, tol=1e-12, scale_dir_cov=False, overrelaxed=False,
                 eigen_method_one=True, max_init_tries=100, chains=None, device=None, max_evals=100000):
        if not np.isinf(m):
            raise NotImplementedError("Slice: only m = inf is supported (the reference's finite-m branch cannot run, slice.py:113-129)")
        MCMCBase.__init__(self, bsmodel, -1, theta=theta, seed=seed, chains=chains, device=device)
        self.w, self.m, self.lower, self.upper = w, m, lower, upper
        self.J, self.l = J, l
        self._tol, self._initscale, self._max_init_tries = tol, initscale, max_init_tries
        self._scale_dir_cov, self._eigen_method_one = scale_dir_cov, eigen_method_one
        self._windowedadaptation = WindowedAdaptation(warmup, windowsize=windowsize, windowscale=windowscale)
        self._sampler = _lib.Sampler(bsmodel.model_id, bsmodel.model_data, self._C, device=self._device, J=J, l=l,
                                     initscale=initscale, warmup=warmup, windowsize=windowsize, windowscale=windowscale,
                                     tol=tol, scale_dir_cov=scale_dir_cov, overrelaxed=False,
                                     eigen_method_one=eigen_method_one, max_init_tries=max_init_tries, variant=3)
        self._sampler.set_slice(w, lower, upper, max_evals)
        self._sampler.seed(self._seeds, theta_given=self._theta_given)

    def _uni_slice(self, rho):
        """slice.py:86-141 from the current theta (also counts the draw, as draw() does before calling it)."""
        self._sampler.slice_step(self._bc(rho, self.D))
        return self._squeeze(self._sampler.get("theta"))

    # the fit / proposal methods of the sinh-arcsinh samplers do not exist on the reference's Slice
    def fit(self, rho):
        raise AttributeError("Slice has no fit()")

    KL = logp_grad_rho = _metropolis_step = fit

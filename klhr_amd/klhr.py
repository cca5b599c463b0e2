"""`KLHR` — KL Hit-and-Run with a 2-parameter Gaussian proposal (klhr.py:15-223), the sibling of KLHRSINH, on MI355X.

Same constructor keywords and defaults as the reference class (note they differ from KLHRSINH's: tol=1e-12,
scale_clip=600, overrelaxed=False, eigen_method_one=True, J capped at D-1).  The device kernels are the same two
mappings as for KLHRSINH with the proposal selected at compile time (`klhr_config.variant = 1`):
  KL(eta, rho)   klhr.py:109-124   y_n = s x_n + m, raw (unclipped) gradient, eta = (m, log s)
  fit(rho)       klhr.py:126-141   1-D mode search, then 2-D BFGS (default gtol) started at (mode, 0.5 log hess_inv)
  MH step        klhr.py:176-195   z' ~ N(m, s) or its over-relaxed version; log q = -log s - z^2/2
  closures       klhr.py:203-214   additionally K <- int(clip(Smoother.optimum(), 1, 50))
"""
import numpy as np

from . import _lib
from .mcmc import MCMCBase
from .windowedadaptation import WindowedAdaptation


class KLHR(MCMCBase):
    def __init__(self, bsmodel, theta=None, seed=None, N=8, K=10, J=2, l=4, initscale=0.1, warmup=1_000, windowsize=50,
                 windowscale=2, tol=1e-12, grad_clip=1e15, scale_clip=600, scale_dir_cov=False, overrelaxed=False,
                 eigen_method_one=True, max_init_tries=100, chains=None, device=None):
        super().__init__(bsmodel, -1, theta=theta, seed=seed, chains=chains, device=device)
        self.N, self._K0, self.l = N, K, l
        self.J = J if J < self.D else self.D - 1            # klhr.py:39
        self._tol, self._grad_clip, self._scale_clip = tol, grad_clip, scale_clip
        self._max_init_tries, self._initscale = max_init_tries, initscale
        self._scale_dir_cov, self._overrelaxed, self._eigen_method_one = scale_dir_cov, overrelaxed, eigen_method_one
        self._windowedadaptation = WindowedAdaptation(warmup, windowsize=windowsize, windowscale=windowscale)
        self._sampler = _lib.Sampler(bsmodel.model_id, bsmodel.model_data, self._C, device=self._device, N=N, K=K, J=self.J, l=l,
                                     initscale=initscale, warmup=warmup, windowsize=windowsize, windowscale=windowscale,
                                     tol=tol, grad_clip=grad_clip, scale_clip=scale_clip, scale_dir_cov=scale_dir_cov,
                                     overrelaxed=overrelaxed, eigen_method_one=eigen_method_one,
                                     max_init_tries=max_init_tries, variant=1)
        self.x, self.w = self._sampler.x.copy(), self._sampler.w.copy()
        self._sampler.seed(self._seeds, theta_given=self._theta_given)

    @property
    def K(self):
        k = self._sampler.get("kstate")[:, 2].astype(int)
        return int(k[0]) if self.chains is None else k

    @property
    def grad_evals(self):
        g = self._sampler.get("grad_evals")
        return int(g[0]) if self.chains is None else g

    @property
    def _draw(self):
        return int(self._sampler.get("counters")[0, 0])

    def _unpack(self, eta):
        c = self._scale_clip
        return eta[0], np.exp(np.clip(eta[1], -c, c))

    def draw(self):
        self._sampler.step(1)
        return self._squeeze(self._sampler.get("theta"))

    def _advance(self, n):
        self._sampler.step(n, record_draws=True)
        return self._sampler.draws(0, n)

    def draws(self, n):
        d = self._advance(n)
        return d[:, 0, :] if self.chains is None else d

    def _eta4(self, eta):
        eta = np.asarray(eta, dtype=np.float64).reshape(-1, 2)
        e4 = np.zeros((self._C, 4))
        e4[:, :2] = eta
        return e4

    def _random_direction(self):
        return self._squeeze(self._sampler.random_direction())

    def fit(self, rho):
        rho = np.broadcast_to(np.asarray(rho, dtype=np.float64).reshape(-1, self.D), (self._C, self.D))
        return self._squeeze(self._sampler.fit(rho)[:, :2])

    def KL(self, eta, rho):
        rho = np.broadcast_to(np.asarray(rho, dtype=np.float64).reshape(-1, self.D), (self._C, self.D))
        v, g = self._sampler.kl_eval(self._eta4(eta), rho)
        return (float(v[0]), g[0, :2]) if self.chains is None else (v, g[:, :2])

    def logp_grad_rho(self, xi, rho):
        rho = np.broadcast_to(np.asarray(rho, dtype=np.float64).reshape(-1, self.D), (self._C, self.D))
        out = self._sampler.line_eval(np.broadcast_to(np.asarray(xi, dtype=np.float64).reshape(-1), (self._C,)), rho)
        return (float(out[0, 0]), float(out[0, 1])) if self.chains is None else (out[:, 0], out[:, 1])

    def _logq(self, x, eta):
        m, s = self._unpack(eta)
        z = (np.asarray(x) - m) / s
        return -np.log(s) - 0.5 * z * z

    def _metropolis_step(self, eta, rho):
        rho = np.broadcast_to(np.asarray(rho, dtype=np.float64).reshape(-1, self.D), (self._C, self.D))
        self._sampler.metropolis(self._eta4(eta), rho)
        return self._squeeze(self._sampler.get("theta"))

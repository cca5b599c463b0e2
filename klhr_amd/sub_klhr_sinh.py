"""`SUBKLHRSINH` — the 3-parameter sinh-arcsinh sibling (sub_klhr_sinh.py:14-284): tail weight d fixed at 1,
eta = (m, log s, e); its `_logp_grad` clips at `grad_clip` instead of `scale_clip` (sub_klhr_sinh.py:151-154) and its
fit draws `rng.normal(size=3)` (sub_klhr_sinh.py:184).  Same kernels as KLHRSINH with `klhr_config.variant = 2`.
(The reference prints the draw counter every 5 000 draws, sub_klhr_sinh.py:257-258; the device path does not.)"""
import numpy as np

from . import _lib
from .klhr_sinh import KLHRSINH
from .mcmc import MCMCBase
from .windowedadaptation import WindowedAdaptation


class SUBKLHRSINH(KLHRSINH):
    def __init__(self, bsmodel, theta=None, seed=None, N=8, K=10, J=2, l=4, initscale=0.1, warmup=1_000, windowsize=50,
                 windowscale=2, tol=1e-10, grad_clip=1e15, scale_clip=300, scale_dir_cov=False, overrelaxed=True,
                 eigen_method_one=False, max_init_tries=100, chains=None, device=None):
        MCMCBase.__init__(self, bsmodel, -1, theta=theta, seed=seed, chains=chains, device=device)
        self.N, self.K, self.J, self.l = N, K, J, l
        self._tol, self._grad_clip, self._scale_clip = tol, grad_clip, scale_clip
        self._max_init_tries, self._initscale = max_init_tries, initscale
        self._scale_dir_cov, self._overrelaxed, self._eigen_method_one = scale_dir_cov, overrelaxed, eigen_method_one
        self._windowedadaptation = WindowedAdaptation(warmup, windowsize=windowsize, windowscale=windowscale)
        self._sampler = _lib.Sampler(bsmodel.model_id, bsmodel.model_data, self._C, device=self._device, N=N, K=K, J=J, l=l,
                                     initscale=initscale, warmup=warmup, windowsize=windowsize, windowscale=windowscale,
                                     tol=tol, grad_clip=grad_clip, scale_clip=scale_clip, scale_dir_cov=scale_dir_cov,
                                     overrelaxed=overrelaxed, eigen_method_one=eigen_method_one,
                                     max_init_tries=max_init_tries, variant=2)
        self.x, self.w = self._sampler.x.copy(), self._sampler.w.copy()
        self._sampler.seed(self._seeds, theta_given=self._theta_given)

    # eta has 3 entries on this class; the C ABI carries it in 4 slots (last = 0)
    def _bc(self, a, cols):
        a = np.asarray(a, dtype=np.float64)
        if cols == 4:
            a3 = a.reshape(-1, 3)
            a = np.concatenate([a3, np.zeros((a3.shape[0], 1))], axis=1)
        return np.broadcast_to(a.reshape(-1, cols), (self._C, cols))

    def fit(self, rho):
        return self._squeeze(self._sampler.fit(KLHRSINH._bc(self, rho, self.D))[:, :3])

    def KL(self, eta, rho):
        v, g = self._sampler.kl_eval(self._bc(eta, 4), KLHRSINH._bc(self, rho, self.D))
        return (float(v[0]), g[0, :3]) if self.chains is None else (v, g[:, :3])

    def _unpack(self, eta):
        c = self._scale_clip
        return eta[0], np.exp(np.clip(eta[1], -c, c)) + self._tol, eta[2]

    def _prim(self, x, eta):
        x = np.atleast_1d(np.asarray(x, dtype=np.float64))
        e3 = np.asarray(eta, dtype=np.float64).reshape(-1, 3)
        E = np.broadcast_to(np.concatenate([e3, np.zeros((e3.shape[0], 1))], axis=1), (x.size, 4))
        return self._sampler.primitives(x, E), np.ndim(x) == 0 or x.size == 1

    def _grad_T(self, x, eta):
        o, _ = self._prim(x, eta)
        return o[0, 4:7]

    def _grad_log_abs_jac(self, x, eta):
        o, _ = self._prim(x, eta)
        return o[0, 8:11]

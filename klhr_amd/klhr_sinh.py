"""`KLHRSINH` — KL Hit-and-Run with a sinh-arcsinh proposal (klhr_sinh.py:14-289) on MI355X.

Same constructor keywords, attributes and methods as the reference class; the work is done by the HIP library
(include/klhr.h) for `chains` independent chains at once.  Method -> C ABI:
    draw()                    klhr_step(1)              klhr_sinh.py:262-289
    sample(M)                 klhr_step(M-1, record)    mcmc.py:31-37
    fit(rho)                  klhr_fit                  klhr_sinh.py:182-201
    KL(eta, rho)              klhr_kl_eval              klhr_sinh.py:163-176
    logp_grad_rho(t, rho)     klhr_line_eval            klhr_sinh.py:178-180
    _random_direction()       klhr_random_direction     klhr_sinh.py:203-213
    _metropolis_step(eta,rho) klhr_metropolis           klhr_sinh.py:242-260
    _T/_T_inv/_log_q/...      klhr_primitives           klhr_sinh.py:112-156,230-240
`overrelaxed=True` (the reference's constructor default; its experiment CLIs pass False unless -o) is supported: the
reference draws the binomial/beta variates of the over-relaxed proposal from SciPy's unseeded global RNG, here they
come from the chain's own PCG64 stream (same distributions, documented consumption; parity is statistical).
"""
import numpy as np

from . import _lib
from .mcmc import MCMCBase
from .windowedadaptation import WindowedAdaptation


class KLHRSINH(MCMCBase):
    def __init__(self, bsmodel, theta=None, seed=None, N=8, K=10, J=2, l=4, initscale=0.1, warmup=1_000, windowsize=50,
                 windowscale=2, tol=1e-10, grad_clip=1e15, scale_clip=300, scale_dir_cov=False, overrelaxed=True,
                 eigen_method_one=False, max_init_tries=100, chains=None, device=None):
        super().__init__(bsmodel, -1, theta=theta, seed=seed, chains=chains, device=device)
        self.N, self.K, self.J, self.l = N, K, J, l
        self._tol, self._grad_clip, self._scale_clip = tol, grad_clip, scale_clip
        self._max_init_tries, self._initscale = max_init_tries, initscale
        self._scale_dir_cov, self._overrelaxed, self._eigen_method_one = scale_dir_cov, overrelaxed, eigen_method_one
        self._windowedadaptation = WindowedAdaptation(warmup, windowsize=windowsize, windowscale=windowscale)
        self._sampler = _lib.Sampler(bsmodel.model_id, bsmodel.model_data, self._C, device=self._device, N=N, K=K, J=J, l=l,
                                     initscale=initscale, warmup=warmup, windowsize=windowsize, windowscale=windowscale,
                                     tol=tol, grad_clip=grad_clip, scale_clip=scale_clip, scale_dir_cov=scale_dir_cov,
                                     overrelaxed=overrelaxed, eigen_method_one=eigen_method_one, max_init_tries=max_init_tries)
        self.x, self.w = self._sampler.x.copy(), self._sampler.w.copy()
        self._sampler.seed(self._seeds, theta_given=self._theta_given)     # MCMCBase.__init__ + _initialize, on device

    # ---- counters the experiment scripts read
    @property
    def grad_evals(self):
        g = self._sampler.get("grad_evals")
        return int(g[0]) if self.chains is None else g

    @property
    def _draw(self):
        return int(self._sampler.get("counters")[0, 0])

    @property
    def _mean(self):
        return self._squeeze(self._sampler.get("mean"))

    @property
    def _cov(self):
        return self._squeeze(self._sampler.get("cov"))

    @property
    def _eigvecs(self):
        return self._squeeze(self._sampler.get("eigvecs"))

    @property
    def _eigvals(self):
        return self._squeeze(self._sampler.get("eigvals"))

    # ---- the hot path
    def draw(self):
        self._sampler.step(1)
        return self._squeeze(self._sampler.get("theta"))

    def _advance(self, n):
        self._sampler.step(n, record_draws=True)
        return self._sampler.draws(0, n)

    def draws(self, n):
        """n draws of every chain in one device call: (n, D) or (n, C, D).  (experiment_funnel.py:49-52 in one launch)"""
        d = self._advance(n)
        return d[:, 0, :] if self.chains is None else d

    # ---- pieces, with the reference's names
    def _bc(self, a, cols):
        a = np.asarray(a, dtype=np.float64)
        return np.broadcast_to(a.reshape(-1, cols), (self._C, cols))

    def _random_direction(self):
        return self._squeeze(self._sampler.random_direction())

    def fit(self, rho):
        return self._squeeze(self._sampler.fit(self._bc(rho, self.D)))

    def KL(self, eta, rho):
        v, g = self._sampler.kl_eval(self._bc(eta, 4), self._bc(rho, self.D))
        return (float(v[0]), g[0]) if self.chains is None else (v, g)

    def logp_grad_rho(self, xi, rho):
        out = self._sampler.line_eval(np.broadcast_to(np.asarray(xi, dtype=np.float64).reshape(-1), (self._C,)),
                                      self._bc(rho, self.D))
        return (float(out[0, 0]), float(out[0, 1])) if self.chains is None else (out[:, 0], out[:, 1])

    def _metropolis_step(self, eta, rho):
        self._sampler.metropolis(self._bc(eta, 4), self._bc(rho, self.D))
        return self._squeeze(self._sampler.get("theta"))

    def _prim(self, x, eta):
        x = np.atleast_1d(np.asarray(x, dtype=np.float64))
        eta = np.asarray(eta, dtype=np.float64)
        E = np.broadcast_to(eta.reshape(-1, 4), (x.size, 4))
        return self._sampler.primitives(x, E), np.ndim(x) == 0 or x.size == 1

    def _unpack(self, eta):
        c = self._scale_clip
        return eta[0], np.exp(np.clip(eta[1], -c, c)) + self._tol, np.exp(np.clip(eta[2], -c, c)) + self._tol, eta[3]

    def _T(self, x, eta):
        o, scalar = self._prim(x, eta)
        return o[0, 0] if scalar and np.ndim(x) == 0 else o[:, 0]

    def _T_inv(self, x, eta):
        o, scalar = self._prim(x, eta)
        return o[0, 1] if scalar and np.ndim(x) == 0 else o[:, 1]

    def _log_abs_jac(self, x, eta):
        o, scalar = self._prim(x, eta)
        return o[0, 2] if scalar and np.ndim(x) == 0 else o[:, 2]

    def _log_q(self, x, eta):
        o, scalar = self._prim(x, eta)
        return o[0, 3] if scalar and np.ndim(x) == 0 else o[:, 3]

    def _grad_T(self, x, eta):
        o, _ = self._prim(x, eta)
        return o[0, 4:8]

    def _grad_log_abs_jac(self, x, eta):
        o, _ = self._prim(x, eta)
        return o[0, 8:12]

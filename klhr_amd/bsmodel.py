"""Log-density plugin surface of the reference (`BSModel`, bsmodel.py:5-55) over built-in device models.

The reference wraps `bridgestan.StanModel`; a Stan program cannot be called from a HIP kernel, so here the `.stan`
file's *stem* selects one of the hand-written device log densities (klhr_amd/csrc/device_models.hpp) and the JSON
data file is packed into the flat fp64 buffer the C ABI takes (include/klhr.h, `enum klhr_model`).  The Python methods
keep the reference's names, argument meaning and error convention (failure -> -inf and a zero gradient,
bsmodel.py:15-30) and are evaluated ON THE GPU through `klhr_log_density_gradient` — there is no CPU path.
"""
import json
from pathlib import Path

import numpy as np

from . import _lib

_STEMS = {"funnel": "funnel", "funnel-100": "funnel", "ill-normal": "ill-normal", "glmm-poisson": "glmm-poisson",
          "normal": "normal", "normal-100": "normal", "ar1": "ar1", "earnings": "earnings",
          "one_normal": "normal"}        # stan/one_normal.stan (slice.py:193): `real y; y ~ normal(0, 1)` = normal with D = 1


def pack_model_data(stem, data):
    """JSON data dict -> flat fp64 buffer (layout documented in include/klhr.h)."""
    if stem in ("funnel", "ill-normal", "normal"):
        return np.array([float(data["D"])])
    if stem == "glmm-poisson":
        year = np.asarray(data["year"], dtype=np.float64)
        counts = np.asarray(data["C"], dtype=np.float64)
        if year.size != int(data["n"]) or counts.size != int(data["n"]):
            raise ValueError("glmm-poisson data: len(year), len(C) must equal n")
        return np.concatenate([[float(data["n"])], year, counts])
    if stem == "ar1":
        return np.array([float(data["N"])])
    if stem == "earnings":
        earn, height = np.asarray(data["earn"], dtype=np.float64), np.asarray(data["height"], dtype=np.float64)
        if earn.size != int(data["N"]) or height.size != int(data["N"]):
            raise ValueError("earnings data: len(earn), len(height) must equal N")
        return np.concatenate([[float(data["N"])], earn, height])
    raise KeyError(stem)


class BSModel:
    def __init__(self, stan_file="", data_file="", stepsize=1.0, warn=False, device=0, data=None):
        self._stan_file = stan_file
        self._data_file = data_file
        stem = Path(str(stan_file)).stem
        if stem not in _STEMS:
            raise ValueError(f"no device log density for Stan program '{stem}'; built in: {sorted(set(_STEMS.values()))}")
        self.stem = _STEMS[stem]
        if data is None:
            with open(data_file) as f:
                data = json.load(f)
        if stem == "one_normal":
            data = dict(data, D=1)
        self.data = data
        self.model_id = _lib.MODEL_IDS[self.stem]
        self.model_data = pack_model_data(self.stem, data)
        self.device = device
        self._D = {"funnel": lambda: int(data["D"]) + 1, "ill-normal": lambda: int(data["D"]),
                   "normal": lambda: int(data["D"]), "glmm-poisson": lambda: int(data["n"]) + 5,
                   "ar1": lambda: int(data["N"]), "earnings": lambda: 4}[self.stem]()
        self._probe = None          # a 1-chain handle used for host-side evaluations

    # -- device plumbing
    def _handle(self):
        if self._probe is None:
            self._probe = _lib.Sampler(self.model_id, self.model_data, 1, device=self.device)
        return self._probe

    def dim(self):
        return self._D

    # -- bsmodel.py:15-37
    def log_density(self, theta, **kws):
        theta = np.asarray(theta, dtype=np.float64)
        try:
            lp, _ = self._handle().log_density_gradient(theta.reshape(-1, self._D), grad=False)
        except _lib.KlhrError:
            raise
        out = np.where(np.isnan(lp), -np.inf, lp)
        return float(out[0]) if theta.ndim == 1 else out

    def log_density_gradient(self, theta, **kws):
        theta = np.asarray(theta, dtype=np.float64)
        lp, g = self._handle().log_density_gradient(theta.reshape(-1, self._D), grad=True)
        bad = np.isnan(lp)
        lp = np.where(bad, -np.inf, lp)
        g[bad] = 0.0
        if theta.ndim == 1:
            return float(lp[0]), g[0]
        return lp, g

    def log_density_hessian(self, theta, **kws):
        """(lp, H): H by central differences of the device gradient (2D gradient evaluations in one batch)."""
        theta = np.asarray(theta, dtype=np.float64)
        D = self._D
        h = 1e-5 * np.maximum(1.0, np.abs(theta))
        pts = np.concatenate([theta + np.diag(h), theta - np.diag(h), theta[None, :]])
        lp, g = self._handle().log_density_gradient(pts, grad=True)
        if not np.isfinite(lp[-1]):
            return -np.inf, np.zeros((D, D))
        H = (g[:D] - g[D:2 * D]) / (2 * h)[:, None]
        return float(lp[-1]), 0.5 * (H + H.T)

    def Hamiltonian(self, theta, rho):
        return -self.log_density(theta) + 0.5 * np.dot(rho, rho)

    # -- constrained <-> unconstrained (bsmodel.py:47-52): only glmm-poisson has a non-identity transform
    def constrain(self, theta):
        theta = np.array(theta, dtype=np.float64)
        if self.stem == "glmm-poisson":
            sigma = np.exp(theta[4])
            theta[4] = sigma
            theta[5:] = sigma * theta[5:]      # vector<multiplier=sigma> eps
        if self.stem == "earnings":
            theta[2:4] = np.exp(theta[2:4])    # real<lower=0> sigma, s
        return theta

    def unconstrain(self, theta):
        theta = np.array(theta, dtype=np.float64)
        if self.stem == "glmm-poisson":
            sigma = theta[4]
            theta[5:] = theta[5:] / sigma
            theta[4] = np.log(sigma)
        if self.stem == "earnings":
            theta[2:4] = np.log(theta[2:4])
        return theta

    def parameter_names(self):
        if self.stem == "funnel":
            return ["double_log_sigma"] + [f"alpha.{i + 1}" for i in range(self._D - 1)]
        if self.stem == "glmm-poisson":
            return ["alpha", "beta.1", "beta.2", "beta.3", "sigma"] + [f"eps.{i + 1}" for i in range(self._D - 5)]
        if self.stem == "earnings":
            return ["beta.1", "beta.2", "sigma", "s"]
        return [f"y.{i + 1}" for i in range(self._D)]

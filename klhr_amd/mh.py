"""`MH` — isotropic random-walk Metropolis comparator (mh.py:7-37) on the device models, C chains at once.
draw() -> klhr_mh_step(1); sample(M) -> klhr_mh_step(M-1, record)."""
import numpy as np

from . import _lib
from .mcmc import MCMCBase


class MH(MCMCBase):
    def __init__(self, model, stepsize, seed=None, theta=None, chains=None, device=None):
        super().__init__(model, stepsize, seed=seed, theta=theta, chains=chains, device=device)
        self._sampler = _lib.Sampler(model.model_id, model.model_data, self._C, device=self._device)
        self._sampler.mh_seed(self._seeds)          # rng + theta = rng.normal(scale=0.1, size=D)   (mcmc.py:9-15)
        if theta is not None:
            self._sampler.set("theta", np.asarray(theta, dtype=np.float64).reshape(-1, self.D))
            self._sampler.set("rng", np.array([_lib.seed_state(s) for s in self._seeds]))   # no normals consumed

    @property
    def _draw(self):
        return int(self._sampler.get("counters")[0, 0])

    def proposal_density(self, thetap, theta):
        z = (np.asarray(thetap) - np.asarray(theta)) / self.stepsize
        return -0.5 * np.sum(z * z, axis=-1)

    def draw(self):
        self._sampler.mh_step(self.stepsize, 1)
        return self._squeeze(self._sampler.get("theta"))

    def _advance(self, n):
        self._sampler.mh_step(self.stepsize, n, record_draws=True)
        return self._sampler.draws(0, n)

    def draws(self, n):
        d = self._advance(n)
        return d[:, 0, :] if self.chains is None else d

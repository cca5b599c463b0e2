"""`MCMCBase` (mcmc.py:3-46) for device-resident chains.

Differences from the reference, all additive:
  * `chains=C` runs C independent chains on the GPU; chain c is seeded with `np.random.default_rng(seed + c)`
    (or with `seed[c]` when a sequence is given).  With `chains=None` the object behaves like the reference's
    single chain: `draw()` returns shape (D,), `sample(M)` shape (M, D).
  * the generator lives on the device; `.rng` returns a NumPy Generator positioned where chain 0's stream is.
"""
import numpy as np

from . import _lib


def chain_seeds(seed, chains):
    if seed is None:
        ss = np.random.SeedSequence()
        return ss.generate_state(chains, dtype=np.uint64)
    if np.ndim(seed) == 0:
        return (np.uint64(int(seed)) + np.arange(chains, dtype=np.uint64)).astype(np.uint64)
    s = np.asarray(seed, dtype=np.uint64)
    if s.shape != (chains,):
        raise ValueError("seed must be a scalar or have one entry per chain")
    return s


class MCMCBase:
    def __init__(self, model, stepsize, theta=None, seed=None, chains=None, device=None):
        self.model = model
        self.D = self.model.dim()
        self.stepsize = stepsize
        self.chains = chains
        self._C = 1 if chains is None else int(chains)
        self._device = getattr(model, "device", 0) if device is None else device
        self._seeds = chain_seeds(seed, self._C)
        self._theta_given = theta is not None
        self._sampler = None       # _lib.Sampler, created by the subclass

    # ---- shape helpers: single-chain objects expose reference shapes
    def _squeeze(self, a):
        return a[0] if self.chains is None else a

    @property
    def theta(self):
        return self._squeeze(self._sampler.get("theta"))

    @theta.setter
    def theta(self, value):
        self._sampler.set("theta", np.asarray(value, dtype=np.float64).reshape(-1, self.D))

    @property
    def rng(self):
        """A NumPy Generator at chain 0's current stream position (read-only view of the device generator)."""
        st = self._sampler.get("rng")[0]
        g = np.random.Generator(np.random.PCG64(0))
        state = g.bit_generator.state
        state["state"] = {"state": (int(st[0]) << 64) | int(st[1]), "inc": (int(st[2]) << 64) | int(st[3])}
        state["has_uint32"], state["uinteger"] = 0, 0
        g.bit_generator.state = state
        return g

    @property
    def acceptance_probability(self):
        a = self._sampler.get("acc")
        return a if self.chains is not None else a[:1]

    def __iter__(self):
        return self

    def __next__(self):
        return self.draw()

    def log_density(self, theta):
        return self.model.log_density(theta)

    def draw(self):
        raise NotImplementedError

    def _advance(self, n):
        raise NotImplementedError

    def sample(self, M):
        """(M, D) [or (M, C, D)]: row 0 is the current point, rows 1.. are M-1 draws (mcmc.py:31-37)."""
        first = self._sampler.get("theta")
        out = np.empty((M, self._C, self.D))
        out[0] = first
        if M > 1:
            out[1:] = self._advance(M - 1)
        return out[:, 0, :] if self.chains is None else out

    def sample_constrained(self, M):
        dr = self.sample(M)
        flat = dr.reshape(-1, self.D)
        return np.array([self.model.constrain(t) for t in flat]).reshape(dr.shape)

"""NumPy stand-in log-density models (TEST INFRASTRUCTURE — never imported by the product path).

The reference evaluates log densities through BridgeStan (bsmodel.py:10-30), a
third-party ctypes->Stan C++ library that is absent from this image (no network,
version unpinned by the reference).  These classes restate, from the text of the
reference's Stan programs, the unnormalised (propto=True) + Jacobian-adjusted
(jacobian=True) log densities BridgeStan would return, with the `BSModel` surface
(`dim`, `log_density`, `log_density_gradient`) so the reference's own sampler code
can be driven by them when generating golden vectors (tests/golden/gen_golden.py).

Closed forms (SURVEY.md §8a row a-L):
  funnel        stan/funnel.stan:4-10      theta=(v, alpha_1..alpha_Da)
  ill-normal    stan/ill-normal.stan:4-11  s_i = i/sqrt(D)
  glmm-poisson  stan/glmm-poisson.stan:6-29
  normal        stan/normal.stan:4-8

Parity status at the BridgeStan boundary: UNPINNED (no Stan toolchain here and the
reference holds no fixture for it); gradients are finite-difference checked in
tests/test_oracle_models.py.
"""
import json
import numpy as np


class _Base:
    def dim(self):
        return self.D

    def log_density(self, theta, **kws):
        return self.log_density_gradient(theta)[0]


class Funnel(_Base):
    name = "funnel"

    def __init__(self, Da):
        self.Da = int(Da)          # Stan data D = number of alphas
        self.D = self.Da + 1

    def data(self):
        return np.array([float(self.Da)])

    def log_density_gradient(self, theta, **kws):
        theta = np.asarray(theta, dtype=np.float64)
        v = theta[0]
        a = theta[1:]
        ss = 0.0
        for x in a:
            ss += x * x
        ev = np.exp(-v)
        lp = -v * v / 18.0 - 0.5 * self.Da * v - 0.5 * ev * ss
        g = np.empty_like(theta)
        g[0] = -v / 9.0 - 0.5 * self.Da + 0.5 * ev * ss
        g[1:] = -ev * a
        return lp, g


class IllNormal(_Base):
    name = "ill-normal"

    def __init__(self, D):
        self.D = int(D)
        self.s = np.arange(1, self.D + 1) / np.sqrt(self.D)

    def data(self):
        return np.array([float(self.D)])

    def log_density_gradient(self, theta, **kws):
        theta = np.asarray(theta, dtype=np.float64)
        z = theta / self.s
        lp = 0.0
        for x in z:
            lp += x * x
        return -0.5 * lp, -z / self.s


class Normal(_Base):
    name = "normal"

    def __init__(self, D):
        self.D = int(D)

    def data(self):
        return np.array([float(self.D)])

    def log_density_gradient(self, theta, **kws):
        theta = np.asarray(theta, dtype=np.float64)
        lp = 0.0
        for x in theta:
            lp += x * x
        return -0.5 * lp, -theta


class GlmmPoisson(_Base):
    name = "glmm-poisson"

    def __init__(self, year, C):
        self.year = np.asarray(year, dtype=np.float64)
        self.C = np.asarray(C, dtype=np.float64)
        self.n = self.year.size
        self.D = self.n + 5
        self.X = np.stack([self.year, self.year * self.year,
                           self.year * self.year * self.year], axis=1)

    def data(self):
        return np.concatenate([[float(self.n)], self.year, self.C])

    def log_density_gradient(self, theta, **kws):
        theta = np.asarray(theta, dtype=np.float64)
        n = self.n
        alpha = theta[0]
        beta = theta[1:4]
        ell = theta[4]
        eps = theta[5:]
        sigma = np.exp(ell)
        lp = -alpha * alpha / 50.0
        for b in beta:
            lp -= b * b / 50.0
        lp -= 0.5 * ell * ell
        g = np.zeros_like(theta)
        sr = 0.0
        sxr = np.zeros(3)
        sre = 0.0
        see = 0.0
        lik = 0.0
        for i in range(n):
            eta = alpha + self.X[i, 0] * beta[0] + self.X[i, 1] * beta[1] \
                + self.X[i, 2] * beta[2] + sigma * eps[i]
            mu = np.exp(eta)
            r = self.C[i] - mu
            lik += self.C[i] * eta - mu
            see += eps[i] * eps[i]
            sr += r
            sxr += self.X[i] * r
            sre += r * eps[i]
            g[5 + i] = -eps[i] + sigma * r
        lp += -0.5 * see + lik
        g[0] = -alpha / 25.0 + sr
        g[1:4] = -beta / 25.0 + sxr
        g[4] = -ell + sigma * sre
        return lp, g


def from_files(stem, data):
    """Build a stand-in from a Stan program stem + parsed JSON data dict."""
    if stem == "funnel":
        return Funnel(data["D"])
    if stem == "ill-normal":
        return IllNormal(data["D"])
    if stem == "normal":
        return Normal(data["D"])
    if stem == "glmm-poisson":
        return GlmmPoisson(data["year"], data["C"])
    raise KeyError(stem)


def load(stem, data_file):
    with open(data_file) as f:
        return from_files(stem, json.load(f))

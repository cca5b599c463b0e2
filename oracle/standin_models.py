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


class AR1(_Base):
    """stan/ar1.stan:1-14: y_1 ~ N(0,1), y_i ~ N(alpha y_{i-1}, beta), alpha = 0.9, beta = sqrt(1 - alpha^2)"""
    name = "ar1"

    def __init__(self, N):
        self.D = int(N)
        self.alpha = 0.9
        self.beta = np.sqrt(1 - self.alpha * self.alpha)

    def data(self):
        return np.array([float(self.D)])

    def log_density_gradient(self, theta, **kws):
        y = np.asarray(theta, dtype=np.float64)
        D, a, b = self.D, self.alpha, self.beta
        g = np.zeros(D)
        lp = -0.5 * y[0] * y[0]
        g[0] = -y[0]
        for i in range(1, D):
            z = (y[i] - a * y[i - 1]) / b
            lp -= 0.5 * z * z
            g[i] = -z / b
            g[i - 1] += a * z / b
        return lp, g


class Earnings(_Base):
    """stan/earnings.stan:1-17 on the unconstrained scale theta = (beta_1, beta_2, log sigma, log s); the Gaussian
    likelihood is evaluated through centred sufficient statistics (exact algebra, numerically stable)."""
    name = "earnings"

    def __init__(self, earn, height):
        e = np.asarray(earn, dtype=np.float64)
        h = np.asarray(height, dtype=np.float64)
        self.D = 4
        self.N = float(e.size)
        self.ebar, self.hbar = e.mean(), h.mean()
        ec, hc = e - self.ebar, h - self.hbar
        self.See, self.Seh, self.Shh = float(ec @ ec), float(ec @ hc), float(hc @ hc)
        self._raw = np.concatenate([[self.N], e, h])

    def data(self):
        return self._raw

    def stats(self):
        return np.array([self.N, self.ebar, self.hbar, self.See, self.Seh, self.Shh])

    def log_density_gradient(self, theta, **kws):
        th = np.asarray(theta, dtype=np.float64)
        b1, b2, lsig, ls = th
        sigma, s = np.exp(lsig), np.exp(ls)
        N = self.N
        delta = self.ebar - b1 - b2 * self.hbar
        SS = self.See - 2 * b2 * self.Seh + b2 * b2 * self.Shh + N * delta * delta
        q1, q2 = b1 * b1 / (5 * s * s), b2 * b2 / (5 * s * s)
        lp = -0.01 * s + ls
        lp += -3 * np.log1p(q1) - ls
        lp += -3 * np.log1p(q2) - ls
        lp += -0.1 * sigma + lsig
        lp += -N * lsig - SS / (2 * sigma * sigma)
        g = np.zeros(4)
        i2 = 1 / (sigma * sigma)
        g[0] = -6 * b1 / (5 * s * s + b1 * b1) + N * delta * i2
        g[1] = -6 * b2 / (5 * s * s + b2 * b2) + (self.Seh - b2 * self.Shh + N * delta * self.hbar) * i2
        g[2] = -0.1 * sigma + 1 - N + SS * i2
        g[3] = -0.01 * s + 1 + (6 * q1 / (1 + q1) - 1) + (6 * q2 / (1 + q2) - 1)
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
    if stem == "ar1":
        return AR1(data["N"])
    if stem == "earnings":
        return Earnings(data["earn"], data["height"])
    raise KeyError(stem)


def load(stem, data_file):
    with open(data_file) as f:
        return from_files(stem, json.load(f))

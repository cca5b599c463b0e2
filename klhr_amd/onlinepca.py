"""Candid covariance-free incremental PCA with amnesia `l` (onlinepca.py:3-39; Weng et al., CCIPCA).

Host-side class for analysis code; the sampler's own copy of this recurrence runs on the device
(klhr_kernels.hpp: pca_update).  Note the reference's quirks kept here: weights (n-1-l)/n are negative while
n <= l, and the deflated residual of component i seeds component i+1."""
import numpy as np


class OnlinePCA:
    def __init__(self, D, K=1, l=0, tol=1e-10):
        self.D, self.K, self.l, self.tol = D, K, l, tol
        self.reset()

    def reset(self):
        self.n = 0
        self.v = np.zeros((self.D, self.K))

    def update(self, u):
        resid = np.array(u, dtype=np.float64)
        self.n += 1
        for i in range(min(self.K, self.n)):
            if i + 1 == self.n:
                self.v[:, i] = resid
                continue
            keep = (self.n - 1 - self.l) / self.n
            col = self.v[:, i]
            proj = resid.dot(col)
            col = keep * col + (1 - keep) * resid * proj / (np.sqrt(col.dot(col)) + self.tol)
            self.v[:, i] = col
            resid = resid - resid.dot(col) * col / (col.dot(col) + self.tol)

    def values(self):
        nv = np.sqrt(np.sum(self.v * self.v, axis=0))
        if not np.all(np.isfinite(nv)):
            nv = np.zeros(self.K)
        return nv + self.tol

    def vectors(self):
        return self.v / self.values()

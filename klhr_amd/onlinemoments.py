"""Streaming mean/variance with the reference's update rule (onlinemoments.py:3-28), batched over leading axes.

`OnlineMoments(D)` is the drop-in used by the experiment harnesses on the host (experiment_accuracy.py:72); inside the
sampler the same recurrence runs on the device (klhr_kernels.hpp: om_update_strided).  `update` accepts any array
whose last axis is D, so one object can track all chains: OnlineMoments(D, batch=(C,))."""
import numpy as np


class OnlineMoments:
    def __init__(self, D, batch=()):
        self.D = D
        self._shape = tuple(batch) + (D,)
        self.reset()

    def reset(self):
        self.N = 0
        self.m = np.zeros(self._shape)
        self.v = np.zeros(self._shape)

    def update(self, x):
        self.N += 1
        w = 1 / self.N
        delta = np.asarray(x, dtype=np.float64) - self.m
        self.m = self.m + delta * w
        self.v = self.v + (-self.v * w + delta * delta * w * (1 - w))

    def mean(self):
        return self.m

    def var(self):
        if self.N <= 2:                      # the reference returns ones until the third update
            return np.ones(self._shape)
        return self.v * self.N / (self.N - 1)

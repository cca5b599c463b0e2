"""Stochastic-approximation tracker (smoother.py:3-20).  Kept for API parity only: in the reference its result is
discarded and its input is never updated (klhr_sinh.py:70,278-287), so the device path does not run it."""


class Smoother:
    def __init__(self, x0, kappa=-0.75):
        self._initial, self._kappa = x0, kappa
        self.reset()

    def reset(self):
        self._count, self._x = 0, self._initial

    def update(self, d):
        self._count += 1
        gain = self._count ** self._kappa
        self._x = gain * (self._x + d) + (1 - gain) * self._x

    def optimum(self):
        return self._x

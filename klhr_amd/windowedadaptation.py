"""Adaptation-window schedule (windowedadaptation.py:1-43): the list of draw indices at which a window closes.
The list itself is computed by the library's host helper `klhr_window_closures`, which is also what the kernels use."""
from . import _lib


class WindowedAdaptation:
    def __init__(self, warmup, windowsize=25, windowscale=2):
        self._warmup, self._windowsize, self._windowscale = warmup, windowsize, windowscale
        self.reset()

    def reset(self):
        self._closures = _lib.window_closures(self._warmup, self._windowsize, self._windowscale)
        self._num_windows = len(self._closures)
        self._idx = 0

    def window_closed(self, m):
        if self._warmup < self._windowsize or not self._closures:
            return False
        hit = m == self._closures[self._idx]
        if hit and self._idx < self._num_windows - 1:
            self._idx += 1
        return hit

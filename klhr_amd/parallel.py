"""Multi-GPU layer: chains shard across the GPUs of a node, one process per GPU (torch.distributed rendezvous).

Parity mode (default) has NO data-path collective: every chain owns its RNG and its own adaptation state, exactly as
one reference process per chain would (run_experiments:27 fans out with GNU parallel).  The optional pooled-adaptation
mode (an extension; NOT reference behaviour) all-reduces, at window closures only, the sufficient
statistics [n, sum theta, sum theta^2] (1 + 2D doubles: latency-bound over xGMI) and overwrites every chain's
direction mean / diagonal covariance with the pooled values.  The collective goes through torch.distributed:
backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.
"""
import os

import numpy as np


def dist_env():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0")))


def shard_range(n_total, rank, world):
    """Contiguous, balanced shard [lo, hi) of n_total units for `rank`."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_seeds(base_seed, n_total, rank, world):
    """Seeds of this rank's chains: chain g (global index) is always seeded with base_seed + g, so the set of chains —
    and every chain's trajectory — is independent of how many GPUs the job runs on."""
    lo, hi = shard_range(n_total, rank, world)
    return (np.uint64(base_seed) + np.arange(lo, hi, dtype=np.uint64)).astype(np.uint64)


def allreduce_sum(vec, dist=None, device=None):
    """Sum a small fp64 host vector over all ranks (RCCL when the process group is nccl, else gloo)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return np.array(vec, dtype=np.float64)
    import torch
    t = torch.from_numpy(np.array(vec, dtype=np.float64))
    if dist.get_backend() == "nccl":
        t = t.to(f"cuda:{device if device is not None else torch.cuda.current_device()}")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def pooled_moments(stats):
    """[n, sum x (D), sum x^2 (D)] -> (mean, unbiased variance) of the pooled window."""
    stats = np.asarray(stats, dtype=np.float64)
    D = (stats.size - 1) // 2
    n = stats[0]
    mean = stats[1:1 + D] / n
    var = (stats[1 + D:] / n - mean * mean) * n / max(n - 1.0, 1.0)
    return mean, np.maximum(var, 0.0)


def pooled_full(stats, D, ncol):
    """[n, sum x (D), sum x x^T (D*D)] -> pooled mean, covariance diagonal, top-`ncol` unit eigenvectors (D, ncol) and
    eigenvalues (ncol, descending) of the pooled covariance.  Eigenvector signs are fixed (largest component positive)
    so that every rank derives bit-identical values from the all-reduced statistics."""
    stats = np.asarray(stats, dtype=np.float64)
    n = stats[0]
    mean = stats[1:1 + D] / n
    S = stats[1 + D:].reshape(D, D) / n - np.outer(mean, mean)
    S = 0.5 * (S + S.T) * n / max(n - 1.0, 1.0)
    w, V = np.linalg.eigh(S)
    order = np.argsort(-w)[:ncol]
    vals = np.maximum(w[order], 0.0)
    vecs = V[:, order]
    sgn = np.sign(vecs[np.argmax(np.abs(vecs), axis=0), np.arange(vecs.shape[1])])
    sgn[sgn == 0] = 1.0
    vecs = vecs * sgn
    if vecs.shape[1] < ncol:                      # D < ncol: pad
        vecs = np.concatenate([vecs, np.zeros((D, ncol - vecs.shape[1]))], axis=1)
        vals = np.concatenate([vals, np.ones(ncol - vals.size)])
    return mean, np.maximum(np.diag(S), 0.0), np.ascontiguousarray(vecs), vals


class ShardedKLHRSINH:
    """All chains of a multi-GPU job behind one object; every rank constructs it with the same arguments."""

    SEGMENT = 50        # pooled="full": draws of at most this many iterations are kept on the device at a time

    def __init__(self, bsmodel, chains_total, seed=0, pooled=False, dist=None, **kw):
        """pooled: False (parity mode, no collective) | True / "diag" (pool mean + per-coordinate variance) |
        "full" (pool the full covariance: fp64 MFMA Gram of the window's draws per GPU, one all-reduce of
        1 + D + D^2 doubles per window closure, eigen-decomposition on every rank)."""
        from .klhr_sinh import KLHRSINH
        self.rank, self.world, self.local_rank = dist_env()
        from . import _lib
        self.local_rank = self.local_rank % max(1, _lib.device_count())      # more ranks than GPUs (rehearsals): share devices
        self.dist = dist
        self.pooled = pooled
        lo, hi = shard_range(chains_total, self.rank, self.world)
        self.lo, self.hi = lo, hi
        self.local = KLHRSINH(bsmodel, seed=shard_seeds(seed, chains_total, self.rank, self.world), chains=hi - lo,
                              device=self.local_rank, overrelaxed=False, **kw)
        self._closures = list(self.local._windowedadaptation._closures)
        self._done = 0
        self._full = pooled == "full"
        self._gram = None

    def advance(self, n, record=False):
        """n iterations on this rank's shard.  Pooled mode stops in front of each window closure, all-reduces the
        closing window's statistics, lets the closure iteration run, then overwrites mean/cov with the pooled values."""
        s = self.local._sampler
        out = []

        def run(k, accumulate=True):
            while k > 0:
                kk = min(k, self.SEGMENT) if self._full else k
                s.step(kk, record_draws=record or self._full)
                if record:
                    out.append(s.draws(0, kk))
                if self._full and accumulate:          # statistics of the open window (the closure draw is excluded)
                    g = s.pool_gram(0, kk)
                    self._gram = g if self._gram is None else self._gram + g
                self._done += kk
                k -= kk

        left = n
        while left > 0:
            nxt = next((c for c in self._closures if c > self._done), None) if self.pooled else None
            if nxt is None:
                run(left)
                break
            before = nxt - 1 - self._done
            if before > 0:
                k = min(left, before)
                run(k)
                left -= k
                continue
            if self._full:
                D = self.local.D
                mine = self._gram if self._gram is not None else np.zeros(1 + D + D * D)
                tot = allreduce_sum(mine, self.dist, self.local_rank)
                run(1, accumulate=False)              # the closure iteration (klhr_sinh.py:268-280)
                left -= 1
                self._gram = None
                if tot[0] > 2 and np.isfinite(tot).all():
                    s.pool_apply_full(*pooled_full(tot, D, s.ncol))
            else:
                tot = allreduce_sum(s.pool_stats(), self.dist, self.local_rank)
                run(1)                                # the closure iteration
                left -= 1
                if tot[0] > 2 and np.isfinite(tot).all():
                    mean, var = pooled_moments(tot)
                    s.pool_apply(mean, var)
        return np.concatenate(out) if record else None

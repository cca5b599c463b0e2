"""Multi-GPU layer: chains shard across the GPUs of a node, one process per GPU (torch.distributed rendezvous).

Parity mode (default) has NO data-path collective: every chain owns its RNG and its own adaptation state, exactly as
one reference process per chain would (run_experiments:27 fans out with GNU parallel).  The optional pooled-adaptation
mode (an extension; NOT reference behaviour) all-reduces, at window closures only, the sufficient
statistics [n, sum theta, sum theta^2] (1 + 2D doubles: latency-bound over xGMI) and overwrites every chain's
direction mean / diagonal covariance with the pooled values.  pooled="full" (the north-star's "pool the covariance across shards") keeps everything on the device and inside the library:
LDS-tiled MFMA Gram of the window's draws -> ONE ncclAllReduce (RCCL over xGMI) of 1 + D + D^2 doubles on the handle's
stream (klhr_comm_init / klhr_pool_allreduce, include/klhr.h) -> one D2H copy -> eigen-decomposition on every rank.  No
PyTorch on that path; torch.distributed (gloo) is only the host channel that hands the communicator id to the ranks, and
the stand-in collective of the CPU tests.
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


def exchange_comm_id(rank, world, dist=None):
    """The 128-byte RCCL communicator id of rank 0 on every rank.  Host plumbing only: torch.distributed (gloo) when a process
    group is given, else a file named by KLHR_COMM_FILE (rank 0 writes it, the others wait for it)."""
    from . import _lib
    if world == 1:
        return _lib.comm_unique_id()
    if dist is not None and dist.is_initialized():
        box = [_lib.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return box[0]
    path = os.environ.get("KLHR_COMM_FILE")
    if not path:
        raise RuntimeError("exchange_comm_id: need a torch.distributed process group or KLHR_COMM_FILE")
    if rank == 0:
        with open(path + ".tmp", "wb") as f:
            f.write(_lib.comm_unique_id())
        os.replace(path + ".tmp", path)
    import time
    for _ in range(6000):
        if os.path.exists(path):
            return open(path, "rb").read()
        time.sleep(0.01)
    raise RuntimeError("exchange_comm_id: timed out waiting for " + path)


def pooled_setup_draws(s, n, rank, world, dist, mode="full", warmup=100, windowsize=50, windowscale=2, segment=50):
    """bench.py --pooled: run `n` draws on the raw sampler `s` with pooled adaptation at the window closures, the collective
    inside the library (RCCL).  Returns what was done, with the mean time of an all-reduce call measured on the host."""
    import time
    from . import _lib
    closures = _lib.window_closures(warmup, windowsize, windowscale)
    D, ncol = s.D, s.ncol
    # RCCL wants one device per rank: a rehearsal with more ranks than GPUs (ranks sharing a device) sums on the host instead
    use_rccl = world <= max(1, _lib.device_count())
    if use_rccl:
        s.comm_init(exchange_comm_id(rank, world, dist), rank, world)
    done, times = 0, []
    s.pool_reset()
    while done < n:
        nxt = next((c for c in closures if c > done), None)
        if nxt is None or nxt - 1 - done >= n - done:
            k = min(n - done, segment)
            s.step(k, record_draws=True)
            done += k
            continue
        before = nxt - 1 - done
        if before > 0:
            k = min(before, segment)
            s.step(k, record_draws=True)
            s.pool_accumulate(0, k)
            done += k
            continue
        s.sync()
        t0 = time.perf_counter()
        s.pool_allreduce()
        tot = s.pool_fetch()
        if not use_rccl:
            tot = allreduce_sum(tot, dist)
        times.append(time.perf_counter() - t0)
        s.step(1)                                  # the closure iteration (klhr_sinh.py:268-280)
        done += 1
        s.pool_reset()
        if tot[0] > 2 and np.isfinite(tot).all():
            if mode == "full":
                s.pool_apply_full(*pooled_full(tot, D, ncol))
            else:
                mean = tot[1:1 + D] / tot[0]
                var = (np.diag(tot[1 + D:].reshape(D, D)) / tot[0] - mean * mean) * tot[0] / max(tot[0] - 1.0, 1.0)
                s.pool_apply(mean, np.maximum(var, 0.0))
    return {"mode": mode, "ranks": world,
            "collective": "ncclAllReduce (RCCL) on the handle's stream, inside libklhr_hip.so" if use_rccl else
                          "host sum over gloo (rehearsal: ranks share a GPU, RCCL needs one device per rank)",
            "doubles_per_allreduce": 1 + D + D * D, "closures_pooled": len(times),
            "allreduce_plus_fetch_ms": [t * 1e3 for t in times]}


class ShardedKLHRSINH:
    """All chains of a multi-GPU job behind one object; every rank constructs it with the same arguments."""

    SEGMENT = 50        # pooled="full": draws of at most this many iterations are kept on the device at a time

    def __init__(self, bsmodel, chains_total, seed=0, pooled=False, dist=None, rccl=None, **kw):
        """pooled: False (parity mode, no collective) | True / "diag" (pool mean + per-coordinate variance) |
        "full" (pool the full covariance: fp64 MFMA Gram of the window's draws per GPU, one all-reduce of
        1 + D + D^2 doubles per window closure, eigen-decomposition on every rank).
        rccl: pooled="full" only — True: the all-reduce is ncclAllReduce inside the library (default when the process group
        is not gloo-on-CPU-tensors only, i.e. on a multi-GPU node); False: the statistics are fetched and summed through
        `dist` (gloo) on the host — the stand-in used by the CPU tests and by rehearsals where ranks share one GPU."""
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
        self._rccl = bool(rccl) and self._full
        if self._rccl:
            self.local._sampler.comm_init(exchange_comm_id(self.rank, self.world, dist), self.rank, self.world)
        if self._full:
            self.local._sampler.pool_reset()

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
                if self._full and accumulate:          # statistics of the open window (the closure draw is excluded):
                    s.pool_accumulate(0, kk)           # device-resident, no copy
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
                s.pool_allreduce()                    # packs; with a communicator: ONE ncclAllReduce on the device
                tot = s.pool_fetch()
                if not self._rccl:
                    tot = allreduce_sum(tot, self.dist, self.local_rank)       # host stand-in (gloo)
                run(1, accumulate=False)              # the closure iteration (klhr_sinh.py:268-280)
                left -= 1
                s.pool_reset()
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

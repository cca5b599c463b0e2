"""Multi-GPU layer: chains shard across the GPUs of a node, one process per GPU.

Parity mode (default) has NO data-path collective: every chain owns its RNG and its own adaptation state, exactly as
one reference process per chain would (run_experiments:27 fans out with GNU parallel).  The optional pooled-adaptation
modes (an extension; NOT reference behaviour) pool, at window closures only, the window's draws of all chains of
all ranks: LDS-tiled MFMA Gram of the window's draws on every GPU -> ONE ncclAllReduce (RCCL over xGMI) of 1 + D + D^2 doubles on
the handle's stream (klhr_comm_init / klhr_pool_allreduce, include/klhr.h) -> one D2H copy -> every rank derives the same pooled
values: pooled="full" installs mean, covariance diagonal and the top eigenpairs (the north-star's "pool the covariance across
shards"), pooled="diag" mean and per-coordinate variance only.

This module imports neither PyTorch nor any other collective library: the collective is RCCL inside libklhr_hip.so.  A host
channel is needed once, to hand rank 0's 128-byte communicator id to the other ranks — any object with
`broadcast_object_list` (torch.distributed's gloo group in bench.py) or a file (KLHR_COMM_FILE).  Where RCCL cannot run —
CPU tests, rehearsals whose ranks share one GPU (RCCL wants one device per rank) — the caller injects `host_allreduce`, a
function summing a small fp64 vector over the ranks, as a stand-in.
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


def one_device_per_rank():
    """True when every rank of this node has a GPU of its own (LOCAL_WORLD_SIZE <= visible devices): what RCCL needs."""
    from . import _lib
    _, world, _ = dist_env()
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    return local_world <= max(1, _lib.device_count())


def exchange_comm_id(rank, world, dist=None):
    """The 128-byte RCCL communicator id of rank 0 on every rank.  Host plumbing only: `dist.broadcast_object_list` when a
    process group object is given, else a file named by KLHR_COMM_FILE + the job's MASTER_PORT (rank 0 removes a stale file of
    an earlier run before anything else and writes it atomically; the others wait for a file newer than their own start)."""
    from . import _lib
    if world == 1:
        return _lib.comm_unique_id()
    if dist is not None and dist.is_initialized():
        box = [_lib.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return box[0]
    path = os.environ.get("KLHR_COMM_FILE")
    if not path:
        raise RuntimeError("exchange_comm_id: need a process group with broadcast_object_list or KLHR_COMM_FILE")
    import time
    path = f"{path}.{os.environ.get('MASTER_PORT', '0')}"
    t_start = time.time()
    if rank == 0:
        try:
            os.unlink(path)                     # a stale id of an earlier run would hang the other ranks in ncclCommInitRank
        except FileNotFoundError:
            pass
        with open(path + ".tmp", "wb") as f:
            f.write(_lib.comm_unique_id())
        os.replace(path + ".tmp", path)
    for _ in range(6000):
        try:
            if os.path.getmtime(path) >= t_start - 1.0:
                data = open(path, "rb").read()
                if len(data) == 128:
                    return data
        except OSError:
            pass
        time.sleep(0.01)
    raise RuntimeError("exchange_comm_id: timed out waiting for " + path)


def pooled_setup_draws(s, n, rank, world, dist, mode="full", warmup=100, windowsize=50, windowscale=2, segment=50, host_allreduce=None):
    """bench.py --pooled / its N > 1 config-5 shard: run `n` draws on the raw sampler `s` with pooled adaptation at the window
    closures, the collective inside the library (RCCL).  Returns what was done, with the time of every all-reduce + fetch
    measured on the host.  Ranks that share a GPU (rehearsals) cannot form an RCCL communicator: they sum through
    `host_allreduce` and the returned record says so."""
    import time
    from . import _lib
    closures = _lib.window_closures(warmup, windowsize, windowscale)
    D, ncol = s.D, s.ncol
    use_rccl = world == 1 or one_device_per_rank()
    if use_rccl:
        s.comm_init(exchange_comm_id(rank, world, dist), rank, world)
    elif host_allreduce is None:
        raise RuntimeError("pooled_setup_draws: ranks share a device (no RCCL communicator possible) and no host_allreduce was given")
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
            tot = host_allreduce(tot)
        times.append(time.perf_counter() - t0)
        s.step(1)                                  # the closure iteration (klhr_sinh.py:268-280)
        done += 1
        s.pool_reset()
        if tot[0] > 2 and np.isfinite(tot).all():
            if mode == "full":
                s.pool_apply_full(*pooled_full(tot, D, ncol))
            else:
                s.pool_apply(*pooled_diag(tot, D))
    return {"mode": mode, "ranks": world, "comm_world": s.comm_world(), "one_device_per_rank": bool(use_rccl),
            "collective": "ncclAllReduce (RCCL) on the handle's stream, inside libklhr_hip.so" if use_rccl else
                          "host sum (rehearsal: ranks share a GPU, RCCL needs one device per rank)",
            "doubles_per_allreduce": 1 + D + D * D, "closures_pooled": len(times),
            "allreduce_plus_fetch_ms": [t * 1e3 for t in times]}


def pooled_diag(stats, D):
    """[n, sum x (D), sum x x^T (D*D)] -> pooled mean and unbiased per-coordinate variance (the diagonal of pooled_full's matrix)"""
    stats = np.asarray(stats, dtype=np.float64)
    n = stats[0]
    mean = stats[1:1 + D] / n
    var = (np.diag(stats[1 + D:].reshape(D, D)) / n - mean * mean) * n / max(n - 1.0, 1.0)
    return mean, np.maximum(var, 0.0)


class ShardedKLHRSINH:
    """All chains of a multi-GPU job behind one object; every rank constructs it with the same arguments."""

    SEGMENT = 50        # pooled modes: draws of at most this many iterations are kept on the device at a time

    def __init__(self, bsmodel, chains_total, seed=0, pooled=False, dist=None, rccl=None, host_allreduce=None, **kw):
        """pooled: False (parity mode, no collective) | True / "diag" (pool mean + per-coordinate variance) |
        "full" (pool the full covariance: mean, diagonal, top eigenpairs).  Both pooled modes take the window's statistics from
        the fp64 MFMA Gram of its draws per GPU and ONE all-reduce of 1 + D + D^2 doubles per window closure.
        rccl: None (default) = the all-reduce is ncclAllReduce inside the library whenever there is more than one rank and
        every rank has its own device; True forces a communicator (also with one rank: the 1-GPU test of that path); False
        never creates one.  Without a communicator and with more than one rank, `host_allreduce(vec) -> summed vec` (the
        caller's stand-in: CPU tests, rehearsals whose ranks share a GPU) is required.
        dist: only handed to exchange_comm_id (broadcast_object_list) — no collective runs through it."""
        from .klhr_sinh import KLHRSINH
        self.rank, self.world, self.local_rank = dist_env()
        from . import _lib
        self.local_rank = self.local_rank % max(1, _lib.device_count())      # more ranks than GPUs (rehearsals): share devices
        self.dist = dist
        self.pooled = "diag" if pooled is True else pooled
        lo, hi = shard_range(chains_total, self.rank, self.world)
        self.lo, self.hi = lo, hi
        self.local = KLHRSINH(bsmodel, seed=shard_seeds(seed, chains_total, self.rank, self.world), chains=hi - lo,
                              device=self.local_rank, overrelaxed=False, **kw)
        self._closures = list(self.local._windowedadaptation._closures)
        self._done = 0
        self._full = self.pooled == "full"
        if rccl is None:
            rccl = bool(self.pooled) and self.world > 1 and one_device_per_rank()
        self._rccl = bool(rccl) and bool(self.pooled)
        self._host_allreduce = host_allreduce
        if self.pooled and not self._rccl and self.world > 1 and host_allreduce is None:
            raise RuntimeError("ShardedKLHRSINH: pooled adaptation over several ranks needs RCCL (one device per rank) or host_allreduce")
        if self._rccl:
            self.local._sampler.comm_init(exchange_comm_id(self.rank, self.world, dist), self.rank, self.world)
        if self.pooled:
            self.local._sampler.pool_reset()

    def advance(self, n, record=False):
        """n iterations on this rank's shard.  Pooled mode stops in front of each window closure, all-reduces the
        closing window's statistics, lets the closure iteration run, then overwrites the direction state with the pooled values."""
        s = self.local._sampler
        out = []

        def run(k, accumulate=True):
            while k > 0:
                kk = min(k, self.SEGMENT) if self.pooled else k
                s.step(kk, record_draws=record or bool(self.pooled))
                if record:
                    out.append(s.draws(0, kk))
                if self.pooled and accumulate:         # statistics of the open window (the closure draw is excluded):
                    s.pool_accumulate(0, kk)           # device-resident, no copy
                self._done += kk
                k -= kk

        left = n
        while left > 0:
            nxt = next((c for c in self._closures if c > self._done), None) if self.pooled else None
            if nxt is None:
                run(left, accumulate=False)
                break
            before = nxt - 1 - self._done
            if before > 0:
                k = min(left, before)
                run(k)
                left -= k
                continue
            D = self.local.D
            s.pool_allreduce()                    # packs; with a communicator: ONE ncclAllReduce on the device
            tot = s.pool_fetch()
            if not self._rccl and self.world > 1:
                tot = self._host_allreduce(tot)    # stand-in (CPU tests, ranks sharing a GPU)
            run(1, accumulate=False)              # the closure iteration (klhr_sinh.py:268-280)
            left -= 1
            s.pool_reset()
            if tot[0] > 2 and np.isfinite(tot).all():
                if self._full:
                    s.pool_apply_full(*pooled_full(tot, D, s.ncol))
                else:
                    s.pool_apply(*pooled_diag(tot, D))
        return np.concatenate(out) if record else None

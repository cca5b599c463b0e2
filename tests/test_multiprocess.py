"""world_size-2 `gloo` tests (CPU) of the N>1 path: shard arithmetic, seeds independent of the GPU count, and the
pooled-adaptation all-reduce + moments.  The GPU kernels are not touched here."""
import os
import subprocess
import sys

import numpy as np

from conftest import ROOT

WORKER = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, os.environ["KLHR_ROOT"])
import torch.distributed as dist
from klhr_amd import parallel as P
import torch
rank, world, _ = P.dist_env()
dist.init_process_group(backend="gloo", rank=rank, world_size=world)
def host_sum(vec):          # the stand-in collective a caller injects where RCCL cannot run (klhr_amd/parallel.py imports no torch)
    t = torch.from_numpy(np.array(vec, dtype=np.float64))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.numpy()
C_total, D = 37, 5
lo, hi = P.shard_range(C_total, rank, world)
seeds = P.shard_seeds(1000, C_total, rank, world)
# synthetic per-chain window statistics: this rank's chains hold N draws each
rng = np.random.default_rng(7)
X = rng.normal(size=(C_total, 11, D)) * np.arange(1, D + 1)
mine = X[lo:hi]
stats = np.concatenate([[mine.shape[0] * mine.shape[1]], mine.sum(axis=(0, 1)), (mine ** 2).sum(axis=(0, 1))])
tot = host_sum(stats)
mean, var = P.pooled_moments(tot)
flat = mine.reshape(-1, D)
gram = np.concatenate([[flat.shape[0]], flat.sum(axis=0), (flat.T @ flat).reshape(-1)])
gtot = host_sum(gram)
fm, fdiag, fvec, fval = P.pooled_full(gtot, D, 2)
dm, dv = P.pooled_diag(gtot, D)
t = host_sum(np.array([float(rank + 1)]))
dist.barrier()
with open(os.path.join(os.environ["KLHR_OUT"], f"result_{rank}.json"), "w") as f:
    json.dump({"rank": rank, "lo": lo, "hi": hi, "seeds": [int(s) for s in seeds], "mean": mean.tolist(),
               "var": var.tolist(), "sum": float(t[0]), "fm": fm.tolist(), "fdiag": fdiag.tolist(), "fvec": fvec.tolist(),
               "fval": fval.tolist(), "dm": dm.tolist(), "dv": dv.tolist()}, f)
dist.destroy_process_group()
'''


def test_two_rank_gloo_sharding_and_pooling(tmp_path):
    import json
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, KLHR_ROOT=ROOT, KLHR_OUT=str(tmp_path), MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29731", str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    res = [json.load(open(tmp_path / f"result_{r}.json")) for r in range(2)]
    assert (res[0]["lo"], res[0]["hi"], res[1]["lo"], res[1]["hi"]) == (0, 19, 19, 37)
    assert res[0]["seeds"] + res[1]["seeds"] == list(range(1000, 1037))      # same chains whatever the GPU count
    X = np.random.default_rng(7).normal(size=(37, 11, 5)) * np.arange(1, 6)
    flat = X.reshape(-1, 5)
    for r in res:
        assert np.allclose(r["mean"], flat.mean(axis=0), rtol=1e-12)
        assert np.allclose(r["var"], flat.var(axis=0, ddof=1), rtol=1e-10)
        assert r["sum"] == 3.0
        S = np.cov(flat.T)
        w, V = np.linalg.eigh(S)
        assert np.allclose(r["fm"], flat.mean(axis=0), rtol=1e-12) and np.allclose(r["fdiag"], np.diag(S), rtol=1e-10)
        assert np.allclose(r["dm"], flat.mean(axis=0), rtol=1e-12) and np.allclose(r["dv"], np.diag(S), rtol=1e-10)
        assert np.allclose(r["fval"], w[::-1][:2], rtol=1e-9)
        assert abs(abs(np.array(r["fvec"])[:, 0] @ V[:, -1]) - 1) < 1e-9
    assert res[0]["fvec"] == res[1]["fvec"]          # every rank derives bit-identical pooled directions


def test_parallel_layer_imports_no_torch():
    """north_star: "no PyTorch" on the product path — the multi-GPU layer's collective is RCCL inside the library; a host
    stand-in is injected by the caller."""
    src = open(os.path.join(ROOT, "klhr_amd", "parallel.py")).read()
    assert "import torch" not in src
    from klhr_amd import parallel as P
    assert not hasattr(P, "allreduce_sum")


def test_comm_id_file_exchange_ignores_a_stale_file(tmp_path, monkeypatch):
    """exchange_comm_id over KLHR_COMM_FILE: a file left by an earlier run is never handed out (ranks > 0 wait for a fresh one)."""
    import time
    from klhr_amd import parallel as P
    base = str(tmp_path / "id")
    monkeypatch.setenv("KLHR_COMM_FILE", base)
    monkeypatch.setenv("MASTER_PORT", "4711")
    stale = base + ".4711"
    open(stale, "wb").write(b"x" * 128)
    os.utime(stale, (time.time() - 100, time.time() - 100))
    import threading
    got = {}

    def reader():
        got["id"] = P.exchange_comm_id(1, 2, None)
    t = threading.Thread(target=reader)
    t.start()
    time.sleep(0.3)
    assert "id" not in got                      # the stale file was not accepted
    fresh = bytes(range(128))
    open(stale + ".tmp", "wb").write(fresh)
    os.replace(stale + ".tmp", stale)
    t.join(timeout=10)
    assert got.get("id") == fresh


def test_shard_ranges_cover_everything():
    from klhr_amd import parallel as P
    for n in (1, 7, 8, 65536, 2 ** 20 + 3):
        for w in (1, 2, 3, 8):
            r = [P.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n and all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1

"""Per golden chain trace: how closely the device fit follows the oracle, teacher-forced (the statistics tests/test_gpu_parity.py
asserts).  Percentiles of the eta error over converged records with identical evaluation counts, and every outlier by record.
usage (GPU box): [KLHR_KERNEL=mono|split] python tools/parity_stats.py [substring of fixture names]"""
import sys, os, glob, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle.oracle as oracle
from parity_util import oracle_teacher_forced, make_sampler
from klhr_amd._lib import DIAG_COLS as DC
col = {k: i for i, k in enumerate(DC)}
pat = sys.argv[1] if len(sys.argv) > 1 else ""
G = sorted(os.path.basename(p) for p in glob.glob(os.path.join(ROOT, "tests", "golden", "g8_*_w*.npz")) if pat in p)
print("kernel mode:", os.environ.get("KLHR_KERNEL", "auto"))
for fn in G:
    R = oracle_teacher_forced(oracle, fn, max_records=400)
    g, M = R["g"], R["M"]
    s = make_sampler(R["stem"], R["data"], M, warmup=R["warmup"], **R["flags"])
    s.set("theta", g["theta_before"][:M]); s.set("rng", g["state_before"][:M])
    for f in R["fields"]:
        s.set(f, R["before"][f])
    s.set("counters", R["cnt_before"])
    s.random_direction()
    eta = s.fit(g["rho"][:M])
    d = s.get("diag"); od = R["diag_fit"]
    on1 = np.array([x["nfev1"] for x in od]); on2 = np.array([x["nfev2"] for x in od])
    st1 = np.array([x["status1"] for x in od]); st2 = np.array([x["status2"] for x in od]); of2 = np.array([x["fun2"] for x in od])
    same = (d[:, col["nfev1"]] == on1) & (d[:, col["nfev2"]] == on2)
    conv = (st1 == 0) & (st2 == 0)
    err = (np.abs(eta - R["eta"]) / (1 + np.abs(R["eta"]))).max(axis=1)
    sel = same & conv
    e = np.sort(err[sel]) if sel.any() else np.zeros(1)
    q = lambda p: e[min(len(e) - 1, int(p * len(e)))]
    print("%-32s %s M %3d conv %3d same&conv %3d | eta err p50 %.1e p90 %.1e p99 %.1e max %.1e | >1e-9: %d  >1e-6: %d" % (
        fn, s.kernel_name()[:16], M, conv.sum(), sel.sum(), q(0.5), q(0.9), q(0.99), e[-1], (e > 1e-9).sum(), (e > 1e-6).sum()), flush=True)
    for i in np.flatnonzero(sel & (err > 1e-8)):
        print("      record %3d: eta err %.2e  nfev %d/%d  KL dev %.17g ora %.17g  nit2 dev %d ora %d" % (
            i, err[i], on1[i], on2[i], d[i, col["fun2"]], of2[i], d[i, col["nit2"]], od[i]["nit2"]))

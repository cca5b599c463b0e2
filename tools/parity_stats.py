"""Report, per golden chain trace, how closely the device fit agrees with the oracle (same statistics the parity test asserts)."""
import sys, os, glob, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import oracle.oracle as oracle
from parity_util import oracle_teacher_forced, make_sampler
from klhr_amd._lib import DIAG_COLS as DC
col = {k: i for i, k in enumerate(DC)}
G = sorted(os.path.basename(p) for p in glob.glob('/root/repo/tests/golden/g8_*_w*.npz'))
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
    with np.errstate(invalid="ignore"):
        frel = np.abs(d[:, col["fun2"]] - of2) / (1 + np.abs(of2))
    fl = ~conv
    print("%-34s M %3d conv %3d same&conv %3d | maxerr(same&conv) %.1e | failed %3d same %3d frel<1e-5 %.3f frel<1e-9 %.3f err<1e-5 %.3f nan %d" % (
        fn, M, conv.sum(), (same & conv).sum(), err[same & conv].max() if (same & conv).any() else 0, fl.sum(), (same & fl).sum(),
        np.mean(np.nan_to_num(frel[fl], nan=1) < 1e-5) if fl.any() else 1, np.mean(np.nan_to_num(frel[fl], nan=1) < 1e-9) if fl.any() else 1,
        np.mean(err[fl] < 1e-5) if fl.any() else 1, np.isnan(frel).sum()), flush=True)

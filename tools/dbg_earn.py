"""Debug: earnings teacher-forced fit, device vs oracle, failed-fit records."""
import sys, os, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import oracle.oracle as oracle
from parity_util import oracle_teacher_forced, make_sampler
from klhr_amd._lib import DIAG_COLS as DC
fn = sys.argv[1] if len(sys.argv) > 1 else "g8_earnings_s0_w100.npz"
R = oracle_teacher_forced(oracle, fn, max_records=400)
g, M, flags = R["g"], R["M"], R["flags"]
s = make_sampler(R["stem"], R["data"], M, warmup=R["warmup"], **flags)
s.set("theta", g["theta_before"][:M]); s.set("rng", g["state_before"][:M])
for f in R["fields"]:
    s.set(f, R["before"][f])
s.set("counters", R["cnt_before"])
s.random_direction()
eta = s.fit(g["rho"][:M])
d = s.get("diag"); col = {k: i for i, k in enumerate(DC)}
od = R["diag_fit"]
keys = ["x0", "mode", "nfev1", "nit1", "status1", "init2", "init3", "nfev2", "nit2", "status2", "fun2"]
keys = [k for k in keys if k in col]
for m in range(M):
    o = od[m]
    fr = abs(d[m, col["fun2"]] - o["fun2"]) / (1 + abs(o["fun2"]))
    if not (fr < 1e-5):
        print("rec", m, "frel", fr)
        for k in keys:
            print("   %-8s dev %-24r ora %-24r ref %r" % (k, d[m, col[k]], o[k], g[k][m] if k in g.files else None))
        print("   eta dev", eta[m], "\n   eta ora", R["eta"][m], "\n   eta ref", g["eta"][m])
# KL probe at the final etas (device's own and the oracle's) for the mismatching records
bad = [m for m in range(M) if not (abs(d[m, col["fun2"]] - od[m]["fun2"]) / (1 + abs(od[m]["fun2"])) < 1e-5)]
s.set("theta", g["theta_before"][:M])
v_dev, g_dev = s.kl_eval(eta, g["rho"][:M])
v_ora, g_ora = s.kl_eval(R["eta"], g["rho"][:M])
c = oracle.Chain(R["stem"], R["data"], seed=0, warmup=R["warmup"])
for m in bad[:6]:
    c.set("theta", g["theta_before"][m])
    print("rec", m, "probe@dev_eta", v_dev[m], "oracle@dev_eta", c.KL(eta[m], g["rho"][m])[0], "| probe@ora_eta", v_ora[m], "oracle", c.KL(R["eta"][m], g["rho"][m])[0], "fit fun2", d[m, col["fun2"]])

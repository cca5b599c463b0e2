"""Debug: per-pass device trace (debug lib with printf) vs oracle evaluation points for one earnings record."""
import sys, os, ctypes, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import oracle.oracle as oracle
from parity_util import oracle_teacher_forced, make_sampler
fn = "g8_earnings_s0_w100.npz"
rec = int(sys.argv[1]) if len(sys.argv) > 1 else 103
R = oracle_teacher_forced(oracle, fn, max_records=rec + 1)
g = R["g"]
# oracle trace
c = oracle.Chain(R["stem"], R["data"], seed=0, warmup=R["warmup"])
for f in R["fields"]:
    c.set(f, R["before"][f][rec])
c.set("theta", g["theta_before"][rec]); c.set("rng", g["state_before"][rec])
c.random_direction()
buf = np.zeros((4096, 4))
L = oracle.lib(); L.klhr_oracle_trace.argtypes = [ctypes.c_void_p, ctypes.c_long]; L.klhr_oracle_trace_count.restype = ctypes.c_long
L.klhr_oracle_trace(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(4096))
c.fit(g["rho"][rec])
n = L.klhr_oracle_trace_count()
L.klhr_oracle_trace(None, 0)
print("oracle points", n)
for i in range(n):
    if buf[i, 1] != 0 or buf[i, 2] != 0:
        print("ora", i, " ".join("%.17g" % v for v in buf[i]), "KL", c.KL(buf[i].copy(), g["rho"][rec])[0])
sys.stdout.flush()
C = 1
s = make_sampler(R["stem"], R["data"], C, warmup=R["warmup"])
s.set("theta", np.tile(g["theta_before"][rec], (C, 1)))
s.set("rng", np.tile(g["state_before"][rec], (C, 1)))
for f in R["fields"]:
    s.set(f, np.tile(R["before"][f][rec][None], (C,) + (1,) * R["before"][f][rec].ndim))
s.set("counters", np.tile(R["cnt_before"][rec], (C, 1)))
s.random_direction()
eta = s.fit(np.tile(g["rho"][rec], (C, 1)))
print("dev eta", eta)

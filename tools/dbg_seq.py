import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT)
import bench
seq = sys.argv[1:]
cfg = {"f100": ("funnel",100,131072), "ill": ("ill-normal",100,16384), "glmm": ("glmm-poisson",45,8192), "f10": ("funnel",10,65536), "ess": None}
for k in seq:
    if k == "ess":
        print("ess", bench.ess_run(0, 1, 1.0)["estimated_on"]["its_own_it_per_s"], flush=True); continue
    m,d,c = cfg[k]
    r,_ = bench.run_workload(m,d,c,50,0,3,0,1,0,123,None)
    print(k, "%.3e" % r["value"], flush=True)

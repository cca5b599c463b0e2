"""it/s of one configuration over chain counts: python tools/dbg_scale.py model D C1 C2 ..."""
import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT)
import bench
m, d = sys.argv[1], int(sys.argv[2])
for c in map(int, sys.argv[3:]):
    r,_ = bench.run_workload(m,d,c,50,0,3,0,1,0,123,None)
    print(m, d, c, "%.3e" % r["value"], flush=True)

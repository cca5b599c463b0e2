#!/bin/bash
# usage (GPU box): bash tools/size_sweep.sh <model> <dim> "<chain counts>" -> it/s with the node-parallel (KLHR_FITN=1) and the lane = chain (KLHR_FITN=0) fit kernel
R=$GRAFT_REPO_ROOT
for C in $3; do for f in 1 0; do
  KLHR_FITN=$f python3 $R/bench.py --model $1 --dim $2 --chains $C --steps 50 --warmup 20 --no-cpu-baseline --no-ess --no-configs 2>&1 | python3 -c "import sys,json; [print('$1-$2 x $C fitn=$f', '%.4g it/s' % json.loads(l)['value'], {k: round(v['mean_ms'],4) for k,v in json.loads(l)['roofline']['kernels'].items()}) for l in sys.stdin if l.startswith('{')]"
done; done

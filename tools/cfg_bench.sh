#!/bin/bash
# usage (GPU box): bash tools/cfg_bench.sh [funnel100|ill|glmm|f10 ...]  -> it/s and per-kernel times of BASELINE configurations
R=$GRAFT_REPO_ROOT
for cfg in "$@"; do
  case $cfg in
    funnel100) A="--model funnel --dim 100 --chains 131072 --steps 50 --warmup 20";;
    ill) A="--model ill-normal --dim 100 --chains 16384 --steps 50 --warmup 20";;
    glmm) A="--model glmm-poisson --dim 45 --chains 8192 --steps 50 --warmup 20";;
    f10) A="--model funnel --dim 10 --chains 65536 --steps 200 --warmup 50";;
  esac
  python3 $R/bench.py $A --no-cpu-baseline --no-ess --no-configs 2>&1 | python3 -c "import sys,json; [print('$cfg', '%.4g it/s' % json.loads(l)['value'], {k: round(v['mean_ms'],4) for k,v in json.loads(l)['roofline']['kernels'].items()}) for l in sys.stdin if l.startswith('{')]"
done

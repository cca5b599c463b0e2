#!/bin/bash
# usage (GPU box): bash tools/free_bench.sh  -> it/s of glmm-poisson-45 x 8192 on the lock-step and the free-running octet kernel
R=$GRAFT_REPO_ROOT
B="python3 $R/bench.py --model glmm-poisson --dim 45 --chains 8192 --steps 100 --warmup 50 --no-cpu-baseline --no-ess --no-configs"
for mode in octet free; do
  for chunk in 50 100; do
    KLHR_KERNEL=$mode KLHR_CHUNK=$chunk $B 2>&1 | python3 -c "import sys,json; [print('$mode chunk $chunk', '%.4g it/s' % json.loads(l)['value'], json.loads(l)['roofline']['kernels']) for l in sys.stdin if l.startswith('{')]"
  done
done

#!/bin/bash
# usage (GPU box): bash tools/ab_libs.sh <other-library.so> <repeats> -- "<model> <dim> <chains>" ...   -> same-box A/B of the production library against another build
R=$GRAFT_REPO_ROOT
other=$1; reps=$2; shift 3
for cfg in "$@"; do
  set -- $cfg
  for r in $(seq $reps); do
    for lib in prod other; do
      if [ $lib = prod ]; then unset KLHR_LIB; else export KLHR_LIB=$R/$other; fi
      python3 $R/bench.py --model $1 --dim $2 --chains $3 --steps 50 --warmup 20 --no-cpu-baseline --no-ess --no-configs 2>&1 | python3 -c "import sys,json; [print('$lib $1-$2 x $3', '%.4g it/s' % json.loads(l)['value'], {k: round(v['mean_ms'],4) for k,v in json.loads(l)['roofline']['kernels'].items()}) for l in sys.stdin if l.startswith('{')]"
    done
  done
done

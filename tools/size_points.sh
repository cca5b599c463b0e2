#!/bin/bash
# usage (GPU box): bash tools/size_points.sh <model> <dim> <chains...>  -> it/s, us per iteration and per-kernel times at several chain counts
R=$GRAFT_REPO_ROOT
m=$1; d=$2; shift 2
for c in "$@"; do
  python3 $R/bench.py --model $m --dim $d --chains $c --steps 50 --warmup 20 --no-cpu-baseline --no-ess --no-configs 2>&1 | python3 -c "import sys,json; [print('$m-$d x $c', '%.4g it/s' % json.loads(l)['value'], '%.1f us/iter' % (1e3*json.loads(l)['ms_per_step']), {k: round(v['mean_ms'],4) for k,v in json.loads(l)['roofline']['kernels'].items()}) for l in sys.stdin if l.startswith('{')]"
done

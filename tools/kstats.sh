#!/bin/bash
# usage (GPU box): bash tools/kstats.sh <tag> <model> <dim> <chains> [steps] [extra bench flags]: per-kernel times of a bench run
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; model=$2; dim=$3; chains=$4; steps=${5:-50}; shift $(( $# < 5 ? $# : 5 ))
rm -rf $R/gpurun_out/ks_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ks_$tag -- python3 $R/bench.py --model $model --dim $dim --chains $chains --steps $steps --warmup 150 --no-cpu-baseline --no-ess --no-configs "$@" > $R/gpurun_out/ks_$tag.log 2>&1 || { tail -5 $R/gpurun_out/ks_$tag.log; exit 1; }
echo "== $tag: $model-$dim x $chains"
python3 - <<PY
import csv,glob,json
f=glob.glob("$R/gpurun_out/ks_$tag/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:4]:
    print(f"  {r['Name'][:60]:60s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Percentage']}%")
d=json.loads([l for l in open("$R/gpurun_out/ks_$tag.log").read().splitlines() if l.startswith("{")][-1])
print("  it/s %.3e  ms/step %.4f  hbm frac %.3f  grad evals %.1f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("grad_evals_per_iter",0)))
PY

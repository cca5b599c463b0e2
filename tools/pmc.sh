#!/bin/bash
# usage: tools/pmc.sh <tag> "<counters pass1>" "<counters pass2>" ...   (runs bench.py --steps 50 under rocprofv3 --pmc per pass)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
i=0
for ctrs in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${tag}_$i -- python3 $R/bench.py --steps 50 --warmup 100 --no-cpu-baseline --no-ess > $R/gpurun_out/pmc_${tag}_$i.log 2>&1 || exit 1
  echo "pass $i done"
done

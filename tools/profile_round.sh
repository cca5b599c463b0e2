#!/bin/bash
# usage (on the GPU box, via gpurun): bash tools/profile_round.sh <tag>
# kernel-trace stats + separate FETCH_SIZE / WRITE_SIZE PMC passes of the default bench workload
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1
rm -rf $R/gpurun_out/prof_${tag}_*
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag}_kt -- python3 $R/bench.py --no-cpu-baseline --no-ess > $R/gpurun_out/prof_${tag}_kt.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_${tag}_fetch -- python3 $R/bench.py --no-cpu-baseline --no-ess > $R/gpurun_out/prof_${tag}_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_${tag}_write -- python3 $R/bench.py --no-cpu-baseline --no-ess > $R/gpurun_out/prof_${tag}_write.log 2>&1 || exit 1
tail -n 1 $R/gpurun_out/prof_${tag}_kt.log | cut -c1-400

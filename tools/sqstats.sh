#!/bin/bash
# usage (GPU box): bash tools/sqstats.sh <tag> <model> <dim> <chains> [steps]: SQ counters per kernel of a bench run (means per dispatch)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; model=$2; dim=$3; chains=$4; steps=${5:-20}; shift $(( $# < 5 ? $# : 5 ))
rm -rf $R/gpurun_out/sq_${tag}_*
B="python3 $R/bench.py --model $model --dim $dim --chains $chains --steps $steps --warmup 120 --no-cpu-baseline --no-ess --no-configs $@"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d $R/gpurun_out/sq_${tag}_1 -- $B > $R/gpurun_out/sq_${tag}_1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/sq_${tag}_2 -- $B > $R/gpurun_out/sq_${tag}_2.log 2>&1 || exit 1
python3 $R/tools/sqstats_read.py $R/gpurun_out/sq_${tag}

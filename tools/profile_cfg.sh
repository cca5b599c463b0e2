#!/bin/bash
# usage (on the GPU box, via gpurun): bash tools/profile_cfg.sh <tag> <model> <dim> <chains> [steps] [extra bench flags]
# One BASELINE configuration of bench.py under rocprofv3: kernel-trace stats, two SQ counter passes, then FETCH_SIZE
# and WRITE_SIZE in their own passes (the guide's HBM recipe).  Output under gpurun_out/cfg_<tag>_*; summaries are
# copied into profiles/<tag>/ by tools/profile_cfg_collect.py.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; model=$2; dim=$3; chains=$4; steps=${5:-100}; shift $(( $# < 5 ? $# : 5 ))
B="python3 $R/bench.py --model $model --dim $dim --chains $chains --steps $steps --warmup 150 --no-cpu-baseline --no-ess --no-configs $@"
O=$R/gpurun_out/cfg_${tag}
rm -rf ${O}_*
rocprofv3 --kernel-trace --stats --output-format csv -d ${O}_kt -- $B > ${O}_kt.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d ${O}_sq1 -- $B > ${O}_sq1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d ${O}_sq2 -- $B > ${O}_sq2.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d ${O}_fetch -- $B > ${O}_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d ${O}_write -- $B > ${O}_write.log 2>&1 || exit 1
tail -n 1 ${O}_kt.log | cut -c1-600

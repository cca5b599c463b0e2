#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_LDS SQ_INSTS_VMEM --kernel-trace --output-format csv -d $R/gpurun_out/pmc_ph -- python3 $R/tools/pmc_phases.py $1 $2 $3 > $R/gpurun_out/pmc_ph.log 2>&1

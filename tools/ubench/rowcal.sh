#!/bin/bash
# usage (GPU box): bash tools/ubench/rowcal.sh   -> gpurun_out/rowcal_{fetch,write}/... counter CSVs
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/rowcal_*
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/rowcal_fetch -- $R/tools/ubench/rowcal 10 > $R/gpurun_out/rowcal_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/rowcal_write -- $R/tools/ubench/rowcal 10 > $R/gpurun_out/rowcal_write.log 2>&1 || exit 1
tail -1 $R/gpurun_out/rowcal_fetch.log

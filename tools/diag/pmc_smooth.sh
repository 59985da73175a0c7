#!/bin/bash
# memory-path counters of the smoother's batched fp32 forward (tools/diag/smooth_trace.py): which unit the split message kernel waits for
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/r05
bash tools/pmc_passes.sh smooth -- python $R/tools/diag/smooth_trace.py > $R/gpurun_out/r05/pmc_smooth.log 2>&1
grep -A60 "sm_msg_split_kernel<128, 0>" $R/gpurun_out/pmc_smooth/summary.txt | head -64
cp $R/gpurun_out/pmc_smooth/summary.txt $R/gpurun_out/r05/pmc_smoother.txt

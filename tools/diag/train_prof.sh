#!/bin/bash
# kernel table of the explorer's training step (tools/train_bench.py) -- which kernels the 24 ms of a batched optimizer step are made of
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/r05
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trainprof
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/trainprof -o t -- python $R/tools/train_bench.py > $R/gpurun_out/r05/train_prof.log 2>&1
cd $R
python tools/rocprof_summary.py $(find /tmp/trainprof -name "*.db" | head -1) gpurun_out/r05/train_kernel_stats.txt > /dev/null 2>&1
head -45 gpurun_out/r05/train_kernel_stats.txt | cut -c1-175

cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/planner_trace
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/planner_trace -o t -- python $R/tools/diag/planner_chunks.py 1024 > $R/gpurun_out/g22_run.log 2>&1
cd $R
python tools/rocprof_summary.py $(find gpurun_out/planner_trace -name "*.db" | head -1) gpurun_out/g22_planner_kernels.txt > /dev/null 2>&1
find gpurun_out/planner_trace -name "*.db" -delete
head -30 gpurun_out/g22_planner_kernels.txt | cut -c1-170

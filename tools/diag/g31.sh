cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/planner_trace
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/planner_trace -o t -- python $R/tools/diag/planner_one.py 1024 256 1 > $R/gpurun_out/g31_run.log 2>&1
cd $R
python tools/rocprof_summary.py $(find gpurun_out/planner_trace -name "*.db" | head -1) gpurun_out/g31_planner_kernels.txt > /dev/null 2>&1
find gpurun_out/planner_trace -name "*.db" -delete
grep problems gpurun_out/g31_run.log; head -24 gpurun_out/g31_planner_kernels.txt | cut -c1-170

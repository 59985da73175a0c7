cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/sm_trace
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/sm_trace -o t -- python $R/tools/diag/smooth_trace.py smooth_2d_attv3 2 256 fp32 > $R/gpurun_out/g39_run.log 2>&1
cd $R
python tools/rocprof_summary.py $(find gpurun_out/sm_trace -name "*.db" | head -1) gpurun_out/g39_kernels.txt > /dev/null 2>&1
find gpurun_out/sm_trace -name "*.db" -delete
head -12 gpurun_out/g39_kernels.txt | cut -c1-170

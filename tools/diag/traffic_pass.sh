#!/bin/bash
# Run ON THE GPU BOX: FETCH_SIZE / WRITE_SIZE passes of one bench workload and the profiles/kernel_traffic.json entry of
# one kernel:   tools/diag/traffic_pass.sh '<kernel-like>' '<workload key>' <bench args...>
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
KL=$1; WK=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/tp_fetch $R/gpurun_out/tp_write
BA="--steps 3 --warmup 1 --no-cpu-baseline --planner-problems 0 --strong-leg 0 --pcie-steps 0 --dense-steps 0 --bf16x3-steps 0 --single-steps 0 --inflight-steps 0 --other-configs-steps 0"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/tp_fetch -o p -- python $R/bench.py $BA "$@" > $R/gpurun_out/tp_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/tp_write -o p -- python $R/bench.py $BA "$@" > $R/gpurun_out/tp_write.log 2>&1
cd $R
python tools/traffic_json.py --kernel-like "$KL" --workload "$WK" --note "bench.py $BA $*" $(find gpurun_out/tp_fetch gpurun_out/tp_write -name "p_results.db")
mkdir -p gpurun_out/profiles_out && cp profiles/kernel_traffic.json gpurun_out/profiles_out/
find gpurun_out/tp_fetch gpurun_out/tp_write -name "*.db" -delete

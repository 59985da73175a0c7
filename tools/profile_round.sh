#!/bin/bash
# Run ON THE GPU BOX (via gpurun): bench + rocprofv3 kernel trace + separate PMC passes for HBM traffic.
# Outputs go to gpurun_out/; summaries are produced afterwards by tools/rocprof_summary.py / tools/pmc_summary.py.
set -u
R=$PWD
mkdir -p $R/gpurun_out
timeout 900 python bench.py > $R/gpurun_out/bench.json 2> $R/gpurun_out/bench.err
cat $R/gpurun_out/bench.json
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/trace $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write $R/gpurun_out/pmc_sq
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/trace -o t -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --planner-problems 0 --pcie-steps 0 --bf16x3-steps 0 --single-steps 0 --inflight-steps 0 --other-configs-steps 0 > $R/gpurun_out/trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --planner-problems 0 --pcie-steps 0 --bf16x3-steps 0 --single-steps 0 --inflight-steps 0 --other-configs-steps 0 --unique 64 > $R/gpurun_out/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --planner-problems 0 --pcie-steps 0 --bf16x3-steps 0 --single-steps 0 --inflight-steps 0 --other-configs-steps 0 --unique 64 > $R/gpurun_out/pmc_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_sq -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --planner-problems 0 --pcie-steps 0 --bf16x3-steps 0 --single-steps 0 --inflight-steps 0 --other-configs-steps 0 --unique 64 > $R/gpurun_out/pmc_sq.log 2>&1
ls $R/gpurun_out
cd $R
timeout 600 python tools/latency.py > $R/gpurun_out/latency.txt 2>&1
for f in "" "--sparse" "--sparse --gpu-graph" "--device-explore --problems 1024" "--device-explore --device-smooth --problems 1024"; do
  timeout 600 python tools/planner_bench.py $f 2>/dev/null | tail -1
done > $R/gpurun_out/planner_bench.txt
timeout 300 python bench.py --no-cpu-baseline --planner-problems 0 --pcie-steps 0 --dense-steps 0 --bf16x3-steps 0 --single-steps 0 --inflight-steps 0 --other-configs-steps 0 --env kuka7 --nodes 2000 --k1 10 --graphs 64 --mlp-dtype bf16 2>/dev/null | tail -1 > $R/gpurun_out/bench_cfg3_kuka7_bf16.json
timeout 300 python bench.py --no-cpu-baseline --planner-problems 0 --pcie-steps 0 --dense-steps 0 --mlp-dtype bf16x3 2>/dev/null | tail -1 > $R/gpurun_out/bench_cfg2_bf16x3.json
ls $R/gpurun_out

#!/bin/bash
# Run ON THE GPU BOX (via gpurun): everything profiles/r03_* is made from.  Outputs under gpurun_out/r03/.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03
rm -rf $O; mkdir -p $O
cd $R
BA="--no-cpu-baseline --planner-problems 0 --pcie-steps 0 --dense-steps 0 --bf16x3-steps 0 --single-steps 0 --inflight-steps 0 --other-configs-steps 0"
# 1. traffic passes first (bench.py reports roofline.traffic from profiles/kernel_traffic.json when the source hash matches)
bash tools/diag/traffic_pass.sh 'pre_resident_kernel<32, 0, true' 'maze2 N=1000 k1=8 graphs=256 fp32' > $O/traffic_edge_pre.log 2>&1
bash tools/diag/traffic_pass.sh 'mp_fused_kernel<32, 0' 'maze2 N=1000 k1=8 graphs=256 fp32' > $O/traffic_mp_cfg2.log 2>&1
bash tools/diag/traffic_pass.sh 'mp_fused_kernel<64, 1' 'kuka7 N=2000 k1=10 graphs=64 bf16' --env kuka7 --nodes 2000 --k1 10 --graphs 64 --mlp-dtype bf16 > $O/traffic_mp_cfg3.log 2>&1
bash tools/diag/traffic_pass.sh 'mp_fused_kernel<32, 1' 'kuka14 N=5000 k1=16 graphs=32 bf16' --env kuka14 --nodes 5000 --k1 16 --graphs 32 --mlp-dtype bf16 > $O/traffic_mp_cfg5.log 2>&1
cp profiles/kernel_traffic.json $O/
# 2. the bench lines
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
timeout 300 python bench.py $BA --env kuka7 --nodes 2000 --k1 10 --graphs 64 --mlp-dtype bf16 2>/dev/null | tail -1 > $O/bench_cfg3_kuka7_bf16.json
timeout 300 python bench.py $BA --env kuka14 --nodes 5000 --k1 16 --graphs 32 --mlp-dtype bf16 2>/dev/null | tail -1 > $O/bench_cfg5_kuka14_bf16.json
timeout 300 python bench.py --no-cpu-baseline --planner-problems 0 --pcie-steps 0 --dense-steps 0 --single-steps 0 --inflight-steps 0 --other-configs-steps 0 --mlp-dtype bf16x3 2>/dev/null | tail -1 > $O/bench_cfg2_bf16x3.json
# 3. kernel trace of the headline command + SQ counters
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python $R/bench.py --steps 10 --warmup 3 $BA > $O/trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_cfg3 -o t -- python $R/bench.py --steps 10 --warmup 3 $BA --env kuka7 --nodes 2000 --k1 10 --graphs 64 --mlp-dtype bf16 > $O/trace_cfg3.log 2>&1
cd $R
python tools/rocprof_summary.py $(find $O/trace -name "*.db" | head -1) $O/bench_kernel_stats.txt > /dev/null 2>&1
python tools/rocprof_summary.py $(find $O/trace_cfg3 -name "*.db" | head -1) $O/bench_cfg3_kernel_stats.txt > /dev/null 2>&1
PMC_GROUPS="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES;SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAVES;FETCH_SIZE;WRITE_SIZE;GRBM_GUI_ACTIVE" bash tools/pmc_passes.sh r03 -- python $R/bench.py --steps 3 --warmup 1 $BA > $O/pmc.log 2>&1
cp gpurun_out/pmc_r03/summary.txt $O/pmc_counters.txt
find $O -name "*.db" -delete
# 4. parity tables, other configs, planner, training
python tools/parity_report.py fp32 > $O/parity_fp32.txt 2>&1
GNNMP_NODE_F64=0 python tools/parity_report.py fp32 > $O/parity_allfp32.txt 2>&1
python tools/parity_report.py bf16 bf16x3 > $O/parity_bf16.txt 2>&1
timeout 900 python tools/latency.py > $O/latency.txt 2>&1
python tools/mixed_bench.py > $O/cfg4_mixed.txt 2>&1
python tools/planner_parity.py > $O/planner_parity.txt 2>&1
python tools/train_bench.py > $O/train_step.txt 2>&1
for f in "" "--sparse" "--sparse --gpu-graph" "--device-explore --problems 1024" "--device-explore --device-smooth --problems 1024"; do
  timeout 600 python tools/planner_bench.py $f 2>/dev/null | tail -1
done > $O/planner_bench.txt
ls -la $O

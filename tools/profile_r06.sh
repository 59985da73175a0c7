#!/bin/bash
# Run ON THE GPU BOX (via gpurun): everything profiles/r06_* is made from.  Outputs under gpurun_out/r06p/.
# (bench.py now runs W warm-up + K timed + 1 + K stage-profiled steps: the kernel tables drop the first W of the W + 2K + 1 launches.)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r06p
rm -rf $O; mkdir -p $O
cd $R
BA="--no-cpu-baseline --planner-problems 0 --strong-leg 0 --pcie-steps 0 --dense-steps 0 --bf16x3-steps 0 --single-steps 0 --inflight-steps 0 --other-configs-steps 0"
# the committed bench LINES keep the two-batches-in-flight leg; kernel traces and counter passes do not (its overlapped launches would enter the per-launch averages)
BL="--no-cpu-baseline --planner-problems 0 --strong-leg 0 --pcie-steps 0 --dense-steps 0 --bf16x3-steps 0 --single-steps 0 --inflight-steps 20 --other-configs-steps 0"
C3="--env kuka7 --nodes 2000 --k1 10 --graphs 64 --mlp-dtype bf16"
C3F="--env kuka7 --nodes 2000 --k1 10 --graphs 64"
C5="--env kuka14 --nodes 5000 --k1 16 --graphs 32 --mlp-dtype bf16"
# 1. traffic passes first (bench.py reports roofline.traffic from profiles/kernel_traffic.json when the source hash matches)
bash tools/diag/traffic_pass.sh 'pre_resident_kernel<32, 0, true' 'maze2 N=1000 k1=8 graphs=256 fp32' > $O/traffic_edge_pre.log 2>&1
bash tools/diag/traffic_pass.sh 'mp_fused_kernel<32, 0' 'maze2 N=1000 k1=8 graphs=256 fp32' > $O/traffic_mp_cfg2.log 2>&1
bash tools/diag/traffic_pass.sh 'mp_fused_w8_kernel<64, 1' 'kuka7 N=2000 k1=10 graphs=64 bf16' $C3 > $O/traffic_mp_cfg3.log 2>&1
bash tools/diag/traffic_pass.sh 'mp_fused_w8_kernel<64, 0' 'kuka7 N=2000 k1=10 graphs=64 fp32' $C3F > $O/traffic_mp_cfg3f.log 2>&1
bash tools/diag/traffic_pass.sh 'mp_fused_kernel<32, 1' 'kuka14 N=5000 k1=16 graphs=32 bf16' $C5 > $O/traffic_mp_cfg5.log 2>&1
cp profiles/kernel_traffic.json $O/
# 2. kernel traces: stats over the steady-state launches only + the launch-time entries bench.py quotes
cd /tmp && export TMPDIR=/tmp
for t in "trace|" "trace_cfg3|$C3" "trace_cfg3f|$C3F" "trace_cfg5|$C5"; do
  d=${t%%|*}; a=${t#*|}
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/$d -o t -- python $R/bench.py --steps 10 --warmup 3 $BA $a > $O/$d.log 2>&1
done
cd $R
python tools/rocprof_summary.py $(find $O/trace -name "*.db" | head -1) $O/bench_kernel_stats.txt --warmup 3 --steps 21 --launch-json 'pre_resident_kernel<32, 0, true' 'maze2 N=1000 k1=8 graphs=256 fp32' > /dev/null 2>&1
python tools/rocprof_summary.py $(find $O/trace_cfg3 -name "*.db" | head -1) $O/bench_cfg3_kernel_stats.txt --warmup 3 --steps 21 --launch-json 'mp_fused_w8_kernel<64, 1' 'kuka7 N=2000 k1=10 graphs=64 bf16' > /dev/null 2>&1
python tools/rocprof_summary.py $(find $O/trace_cfg3f -name "*.db" | head -1) $O/bench_cfg3f_kernel_stats.txt --warmup 3 --steps 21 --launch-json 'pre_kernel<64, 0, true' 'kuka7 N=2000 k1=10 graphs=64 fp32' > /dev/null 2>&1
python tools/rocprof_summary.py $(find $O/trace_cfg5 -name "*.db" | head -1) $O/bench_cfg5_kernel_stats.txt --warmup 3 --steps 21 --launch-json 'mp_fused_kernel<32, 1' 'kuka14 N=5000 k1=16 graphs=32 bf16' > /dev/null 2>&1
cp profiles/kernel_launch_ms.json $O/
# 3. SQ counters of the four shapes (+ the issue-slot entries of the bf16 edge kernels)
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"
G2="SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
PMC_GROUPS="$G1;$G2;FETCH_SIZE;WRITE_SIZE;GRBM_GUI_ACTIVE" bash tools/pmc_passes.sh r06 -- python $R/bench.py --steps 3 --warmup 1 $BA > $O/pmc.log 2>&1
cp gpurun_out/pmc_r06/summary.txt $O/pmc_counters.txt
PMC_GROUPS="$G1;$G2;FETCH_SIZE;WRITE_SIZE;GRBM_GUI_ACTIVE" bash tools/pmc_passes.sh r06c3 -- python $R/bench.py --steps 3 --warmup 1 $BA $C3 > $O/pmc_cfg3.log 2>&1
cp gpurun_out/pmc_r06c3/summary.txt $O/pmc_cfg3_bf16.txt
PMC_GROUPS="$G1;$G2;FETCH_SIZE;WRITE_SIZE;GRBM_GUI_ACTIVE" bash tools/pmc_passes.sh r06c3f -- python $R/bench.py --steps 3 --warmup 1 $BA $C3F > $O/pmc_cfg3f.log 2>&1
cp gpurun_out/pmc_r06c3f/summary.txt $O/pmc_cfg3_fp32.txt
PMC_GROUPS="$G1;$G2;FETCH_SIZE;WRITE_SIZE;GRBM_GUI_ACTIVE" bash tools/pmc_passes.sh r06c5 -- python $R/bench.py --steps 3 --warmup 1 $BA $C5 > $O/pmc_cfg5.log 2>&1
cp gpurun_out/pmc_r06c5/summary.txt $O/pmc_cfg5_bf16.txt
python tools/issue_json.py $O/pmc_cfg3_bf16.txt --kernel-like 'pre_resident_kernel<64, 1, true' --stage edge_pre --workload 'kuka7 N=2000 k1=10 graphs=64 bf16' --tiles 61296 > $O/issue_cfg3.json 2>&1
python tools/issue_json.py $O/pmc_cfg5_bf16.txt --kernel-like 'pre_resident_kernel<32, 1, true' --stage edge_pre --workload 'kuka14 N=5000 k1=16 graphs=32 bf16' --tiles 132000 > $O/issue_cfg5.json 2>&1
cp profiles/kernel_issue.json $O/ 2>/dev/null
find $O gpurun_out/pmc_r06 gpurun_out/pmc_r06c3 gpurun_out/pmc_r06c3f gpurun_out/pmc_r06c5 -name "*.db" -delete
# 3b. executed matrix-pipe FLOPs per step of the four shapes (SQ_INSTS_VALU_MFMA_MOPS_* x 512) -> profiles/kernel_mfma.json
bash tools/mfma_pass.sh > $O/mfma_pass.log 2>&1
cp profiles/kernel_mfma.json $O/ 2>/dev/null
# 4. the bench lines (they quote the traffic / launch-time / issue entries made above)
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err
timeout 300 python bench.py $BL 2>/dev/null | tail -1 > $O/bench_cfg2_two_in_flight.json
timeout 300 python bench.py $BL $C3 2>/dev/null | tail -1 > $O/bench_cfg3_kuka7_bf16.json
timeout 300 python bench.py $BL $C3F 2>/dev/null | tail -1 > $O/bench_cfg3_kuka7_fp32.json
timeout 300 python bench.py $BL $C5 2>/dev/null | tail -1 > $O/bench_cfg5_kuka14_bf16.json
timeout 300 python bench.py --no-cpu-baseline --planner-problems 0 --strong-leg 0 --pcie-steps 0 --dense-steps 0 --single-steps 0 --inflight-steps 20 --other-configs-steps 0 --mlp-dtype bf16x3 2>/dev/null | tail -1 > $O/bench_cfg2_bf16x3.json
GNNMP_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 $BA --strong-leg 512 2>/dev/null | grep '^{' | tail -1 > $O/bench_selflaunch_2ranks_gloo_one_gpu.json
# 5. per-wave timelines of the message-passing launch (diagnostics build, when present)
if [ -f gnn-motion-planning_amd/libgnnmp_trace.so ]; then
  for a in "kuka7 2000 10 64 bf16" "kuka7 2000 10 64 fp32" "maze2 1000 8 256 fp32" "kuka14 5000 16 32 bf16"; do
    GNNMP_LIB=$R/gnn-motion-planning_amd/libgnnmp_trace.so python tools/diag/mp_trace2.py $a 2>&1 | grep -v "GNNMP_LIB\|amdgpu.ids"
  done > $O/mp_wave_timeline.txt
fi
# 6. parity tables, other configs, mixed set, cfg-5 pipeline, planner, training
python tools/parity_report.py fp32 > $O/parity_fp32.txt 2>&1
python tools/parity_report.py bf16 bf16x3 > $O/parity_bf16.txt 2>&1
timeout 900 python tools/latency.py > $O/latency.txt 2>&1
python tools/mixed_bench.py > $O/cfg4_mixed.txt 2>&1
python tools/cfg5_pipeline.py > $O/cfg5_pipeline.json 2>/dev/null
python tools/train_bench.py > $O/train_step.txt 2>&1
for f in "--device-explore --device-smooth --problems 1024" "--device-eval --problems 1024"; do
  timeout 600 python tools/planner_bench.py $f 2>/dev/null | tail -1
done > $O/planner_bench.txt
# 7. round 6: mixed-set cost sweep, small-batch crossover, drop-in forward span, smoother kernels by batch size
python tools/cost_sweep.py 2>&1 | grep -v amdgpu.ids > $O/cost_sweep.txt
bash tools/diag/throttle_probe.sh > $O/dropin_forward.txt 2>&1
python -m pytest tests/test_dropin_latency_gpu.py -q -s 2>&1 | tail -5 >> $O/dropin_forward.txt
for s in 0 1; do GNNMP_SM_STREAM=$s python tools/diag/sm_stream_sizes.py 2>&1 | grep STREAM; done > $O/smoother_stream_sizes.txt
ls -la $O

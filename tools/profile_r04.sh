#!/bin/bash
# Run ON THE GPU BOX (via gpurun): everything profiles/r04_* is made from.  Outputs under gpurun_out/r04/.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04
rm -rf $O; mkdir -p $O
cd $R
BA="--no-cpu-baseline --planner-problems 0 --pcie-steps 0 --dense-steps 0 --bf16x3-steps 0 --single-steps 0 --inflight-steps 0 --other-configs-steps 0"
C3="--env kuka7 --nodes 2000 --k1 10 --graphs 64 --mlp-dtype bf16"
C5="--env kuka14 --nodes 5000 --k1 16 --graphs 32 --mlp-dtype bf16"
# 1. traffic passes first (bench.py reports roofline.traffic from profiles/kernel_traffic.json when the source hash matches)
bash tools/diag/traffic_pass.sh 'pre_resident_kernel<32, 0, true' 'maze2 N=1000 k1=8 graphs=256 fp32' > $O/traffic_edge_pre.log 2>&1
bash tools/diag/traffic_pass.sh 'mp_fused_kernel<32, 0' 'maze2 N=1000 k1=8 graphs=256 fp32' > $O/traffic_mp_cfg2.log 2>&1
bash tools/diag/traffic_pass.sh 'mp_fused_kernel<64, 1' 'kuka7 N=2000 k1=10 graphs=64 bf16' $C3 > $O/traffic_mp_cfg3.log 2>&1
bash tools/diag/traffic_pass.sh 'mp_fused_kernel<32, 1' 'kuka14 N=5000 k1=16 graphs=32 bf16' $C5 > $O/traffic_mp_cfg5.log 2>&1
cp profiles/kernel_traffic.json $O/
# 2. kernel traces of the three shapes: stats over the TIMED launches only (warm-ups dropped) + the launch-time entries bench.py quotes
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python $R/bench.py --steps 10 --warmup 3 $BA > $O/trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_cfg3 -o t -- python $R/bench.py --steps 10 --warmup 3 $BA $C3 > $O/trace_cfg3.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_cfg5 -o t -- python $R/bench.py --steps 10 --warmup 3 $BA $C5 > $O/trace_cfg5.log 2>&1
cd $R
python tools/rocprof_summary.py $(find $O/trace -name "*.db" | head -1) $O/bench_kernel_stats.txt --warmup 3 --steps 10 --launch-json 'pre_resident_kernel<32, 0, true' 'maze2 N=1000 k1=8 graphs=256 fp32' > /dev/null 2>&1
python tools/rocprof_summary.py $(find $O/trace_cfg3 -name "*.db" | head -1) $O/bench_cfg3_kernel_stats.txt --warmup 3 --steps 10 --launch-json 'mp_fused_kernel<64, 1' 'kuka7 N=2000 k1=10 graphs=64 bf16' > /dev/null 2>&1
python tools/rocprof_summary.py $(find $O/trace_cfg5 -name "*.db" | head -1) $O/bench_cfg5_kernel_stats.txt --warmup 3 --steps 10 --launch-json 'mp_fused_kernel<32, 1' 'kuka14 N=5000 k1=16 graphs=32 bf16' > /dev/null 2>&1
cp profiles/kernel_launch_ms.json $O/
# 3. the bench lines (they quote the traffic and launch-time entries made above)
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
timeout 300 python bench.py $BA $C3 2>/dev/null | tail -1 > $O/bench_cfg3_kuka7_bf16.json
timeout 300 python bench.py $BA $C5 2>/dev/null | tail -1 > $O/bench_cfg5_kuka14_bf16.json
timeout 300 python bench.py --no-cpu-baseline --planner-problems 0 --pcie-steps 0 --dense-steps 0 --single-steps 0 --inflight-steps 0 --other-configs-steps 0 --mlp-dtype bf16x3 2>/dev/null | tail -1 > $O/bench_cfg2_bf16x3.json
GNNMP_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python bench.py $BA --strong 256 2>/dev/null | grep '^{' | tail -1 > $O/bench_strong256_forced_dist.json
# 4. SQ counters of the three shapes
PMC_GROUPS="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES;SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAVES;FETCH_SIZE;WRITE_SIZE;GRBM_GUI_ACTIVE" bash tools/pmc_passes.sh r04 -- python $R/bench.py --steps 3 --warmup 1 $BA > $O/pmc.log 2>&1
cp gpurun_out/pmc_r04/summary.txt $O/pmc_counters.txt
PMC_GROUPS="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES;SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR;FETCH_SIZE;WRITE_SIZE;GRBM_GUI_ACTIVE" bash tools/pmc_passes.sh r04c3 -- python $R/bench.py --steps 3 --warmup 1 $BA $C3 > $O/pmc_cfg3.log 2>&1
cp gpurun_out/pmc_r04c3/summary.txt $O/pmc_cfg3_bf16.txt
PMC_GROUPS="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES;FETCH_SIZE;WRITE_SIZE;GRBM_GUI_ACTIVE" bash tools/pmc_passes.sh r04c5 -- python $R/bench.py --steps 3 --warmup 1 $BA $C5 > $O/pmc_cfg5.log 2>&1
cp gpurun_out/pmc_r04c5/summary.txt $O/pmc_cfg5_bf16.txt
find $O gpurun_out/pmc_r04 gpurun_out/pmc_r04c3 gpurun_out/pmc_r04c5 -name "*.db" -delete
# (diagnostics builds, made beforehand on any machine with hipcc: tools/diag/build_variant.sh trace -DGNNMP_MP_TRACE ;
#  tools/diag/build_variant.sh mztrace -DGNNMP_MAZE_TRACE)
[ -f gnn-motion-planning_amd/libgnnmp_mztrace.so ] && GNNMP_LIB=$R/gnn-motion-planning_amd/libgnnmp_mztrace.so python tools/diag/maze_trace.py 2>&1 | grep -v "GNNMP_LIB\|amdgpu.ids" > $O/maze_explore_phases.txt
# 5. per-wave timelines of the message-passing launch (diagnostics build, when present), stream microbenchmark
if [ -f gnn-motion-planning_amd/libgnnmp_trace.so ]; then
  for a in "kuka7 2000 10 64 bf16" "maze2 1000 8 256 fp32" "kuka14 5000 16 32 bf16"; do
    GNNMP_LIB=$R/gnn-motion-planning_amd/libgnnmp_trace.so python tools/diag/mp_trace.py $a 2>&1 | grep -v "GNNMP_LIB\|amdgpu.ids"
  done > $O/mp_wave_timeline.txt
fi
[ -x tools/microbench/stream_read ] && ./tools/microbench/stream_read > $O/stream_read.txt 2>&1
# 6. parity: per-fixture table, full-size census, other configs, mixed set, cfg-5 pipeline, planner, training
python tools/parity_report.py fp32 > $O/parity_fp32.txt 2>&1
python tools/parity_report.py bf16 bf16x3 > $O/parity_bf16.txt 2>&1
python tools/parity_census.py > $O/parity_census.txt 2>&1
timeout 900 python tools/latency.py > $O/latency.txt 2>&1
python tools/mixed_bench.py > $O/cfg4_mixed.txt 2>&1
python tools/cfg5_pipeline.py > $O/cfg5_pipeline.json 2>/dev/null
python tools/planner_parity.py > $O/planner_parity.txt 2>&1
python tools/train_bench.py > $O/train_step.txt 2>&1
for f in "" "--sparse" "--sparse --gpu-graph" "--device-explore --problems 1024" "--device-explore --device-smooth --problems 1024" "--device-eval --problems 1024"; do
  timeout 600 python tools/planner_bench.py $f 2>/dev/null | tail -1
done > $O/planner_bench.txt
ls -la $O

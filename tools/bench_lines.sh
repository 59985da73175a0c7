#!/bin/bash
# Run ON THE GPU BOX, on its own (a fresh box: counter passes before a bench run leave the clocks in the profiler's state and cost ~2.5 %):
# the bench lines kept under profiles/rNN_bench*.json.  Outputs under gpurun_out/bl/.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/bl; rm -rf $O; mkdir -p $O
BL="--no-cpu-baseline --planner-problems 0 --strong-leg 0 --pcie-steps 0 --dense-steps 0 --bf16x3-steps 0 --single-steps 0 --inflight-steps 20 --other-configs-steps 0"
C3="--env kuka7 --nodes 2000 --k1 10 --graphs 64 --mlp-dtype bf16"
C3F="--env kuka7 --nodes 2000 --k1 10 --graphs 64"
C5="--env kuka14 --nodes 5000 --k1 16 --graphs 32 --mlp-dtype bf16"
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err
timeout 300 python bench.py $BL 2>/dev/null | tail -1 > $O/bench_cfg2_two_in_flight.json
timeout 300 python bench.py $BL $C3 2>/dev/null | tail -1 > $O/bench_cfg3_kuka7_bf16.json
timeout 300 python bench.py $BL $C3F 2>/dev/null | tail -1 > $O/bench_cfg3_kuka7_fp32.json
timeout 300 python bench.py $BL $C5 2>/dev/null | tail -1 > $O/bench_cfg5_kuka14_bf16.json
timeout 300 python bench.py --no-cpu-baseline --planner-problems 0 --strong-leg 0 --pcie-steps 0 --dense-steps 0 --single-steps 0 --inflight-steps 20 --other-configs-steps 0 --mlp-dtype bf16x3 2>/dev/null | tail -1 > $O/bench_cfg2_bf16x3.json
BA="--no-cpu-baseline --planner-problems 0 --pcie-steps 0 --dense-steps 0 --bf16x3-steps 0 --single-steps 0 --inflight-steps 0 --other-configs-steps 0"
GNNMP_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 $BA --strong-leg 512 2>/dev/null | grep '^{' | tail -1 > $O/bench_selflaunch_2ranks_gloo_one_gpu.json
( time GNNMP_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 8 2>/dev/null | grep '^{' | tail -1 > $O/bench_selflaunch_8ranks_gloo_one_gpu.json ) 2> $O/bench_8ranks_time.txt
python tools/mixed_bench.py 2>&1 | grep -v amdgpu.ids > $O/cfg4_mixed.txt
for f in "--device-explore --device-smooth --problems 1024" "--device-eval --problems 1024"; do timeout 600 python tools/planner_bench.py $f 2>/dev/null | tail -1; done > $O/planner_bench.txt
python tools/cfg5_pipeline.py > $O/cfg5_pipeline.json 2>/dev/null
timeout 900 python tools/latency.py 2>&1 | grep -v amdgpu.ids > $O/latency.txt
ls -la $O

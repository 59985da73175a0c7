cd $GRAFT_REPO_ROOT
python tools/diag/planner_one.py 1024 128 2 > gpurun_out/g34.log 2>&1
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --pcie-steps 0 --dense-steps 0 --bf16x3-steps 0 --single-steps 0 --inflight-steps 0 --other-configs-steps 0 --planner-problems 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench no-cpu-baseline:', d['config']['planner']['device_planner']['timing'])" >> gpurun_out/g34.log 2>&1
python bench.py --steps 3 --warmup 1 --pcie-steps 0 --dense-steps 0 --bf16x3-steps 0 --single-steps 0 --inflight-steps 0 --other-configs-steps 0 --planner-problems 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench with cpu-baseline:', d['config']['planner']['device_planner']['timing'])" >> gpurun_out/g34.log 2>&1
python tools/planner_bench.py --device-eval --problems 1024 >> gpurun_out/g34.log 2>&1

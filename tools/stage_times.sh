#!/bin/bash
# per-stage times (HIP events inside bench.py) of the headline workload and of the cfg-3 shape, fp32 and bf16;
# run on the GPU box:  gpurun -- 'bash tools/stage_times.sh'
cd $GRAFT_REPO_ROOT
show() { tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['config']['stage_ms_per_step'], d['config']['result_checksum'])"; }
timeout 300 python bench.py --no-cpu-baseline --planner-problems 0 --pcie-steps 0 --dense-steps 0 --bf16x3-steps 0 --single-steps 0 --inflight-steps 0 --other-configs-steps 0 2>&1 | show
timeout 300 python bench.py --no-cpu-baseline --planner-problems 0 --pcie-steps 0 --dense-steps 0 --bf16x3-steps 0 --single-steps 0 --inflight-steps 0 --other-configs-steps 0 --env kuka7 --nodes 2000 --k1 10 --graphs 64 2>&1 | show
timeout 300 python bench.py --no-cpu-baseline --planner-problems 0 --pcie-steps 0 --dense-steps 0 --bf16x3-steps 0 --single-steps 0 --inflight-steps 0 --other-configs-steps 0 --env kuka7 --nodes 2000 --k1 10 --graphs 64 --mlp-dtype bf16 2>&1 | show

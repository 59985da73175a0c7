#!/bin/bash
# bf16 operand mode against the reference cast to bfloat16 (tests/golden/refbf16_*) + the full-size bf16 tests, with their printed figures
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/s4; mkdir -p $O
timeout 900 python -m pytest tests/test_bf16_anchor.py tests/test_full_size_bf16_gpu.py tests/test_explorer_bf16.py -m gpu -s -q 2>&1 | grep -E "ours - ref32|bf16:|bf16 vs fp32|gpu-bf16 vs emulation|passed|failed|Error" > $O/bf16_anchor.txt
cat $O/bf16_anchor.txt
BA="--no-cpu-baseline --planner-problems 0 --strong-leg 0 --pcie-steps 0 --dense-steps 0 --bf16x3-steps 0 --single-steps 0 --inflight-steps 0 --other-configs-steps 0"
python bench.py $BA --env kuka7 --nodes 2000 --k1 10 --graphs 64 2>/dev/null | tail -1 > $O/b_cfg3f.json
python -c "
import json; d=json.load(open('$O/b_cfg3f.json')); print(d['value'], d['config']['stage_roofline'])"

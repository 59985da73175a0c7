#!/bin/bash
# headline shape (maze 1000-node k=8 x 256, fp32): one bench line with the stage split; run ON THE GPU BOX
python bench.py --no-cpu-baseline --planner-problems 0 --strong-leg 0 --pcie-steps 0 --dense-steps 0 --bf16x3-steps 0 --single-steps 0 --inflight-steps 0 --other-configs-steps 0 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r['config']['stage_ms_per_step'], r['roofline']['frac'], r['config']['result_checksum'])"

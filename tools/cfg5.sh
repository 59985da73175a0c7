#!/bin/bash
# cfg5 shape (kuka14 5000-node k=16 x 32, bf16): one bench line with the stage split; run ON THE GPU BOX
python bench.py --no-cpu-baseline --planner-problems 0 --strong-leg 0 --pcie-steps 0 --dense-steps 0 --bf16x3-steps 0 --single-steps 0 --inflight-steps 0 --other-configs-steps 0 --env kuka14 --nodes 5000 --k1 16 --graphs 32 --mlp-dtype bf16 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r['config']['stage_ms_per_step'], r['config']['result_checksum'])"

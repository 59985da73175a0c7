#!/bin/bash
# Run ON THE GPU BOX: policy stage time against resident workgroups per CU (GNNMP_WGS_PER_CU) at the two bf16 shapes and cfg 2
R=${GRAFT_REPO_ROOT:-$PWD}
BA="--steps 20 --warmup 5 --no-cpu-baseline --planner-problems 0 --strong-leg 0 --pcie-steps 0 --dense-steps 0 --bf16x3-steps 0 --single-steps 0 --inflight-steps 0 --other-configs-steps 0"
for cfg in "--env kuka7 --nodes 2000 --k1 10 --graphs 64 --mlp-dtype bf16" "--env kuka14 --nodes 5000 --k1 16 --graphs 32 --mlp-dtype bf16" ""; do
  for w in 0 2 3 4 5 6 8; do
    if [ $w = 0 ]; then unset GNNMP_WGS_PER_CU; else export GNNMP_WGS_PER_CU=$w; fi
    python $R/bench.py $BA $cfg 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-60s wgs/cu=$w  step %.4f ms  policy %.4f  checksum %s' % ('$cfg', d['ms_per_step'], d['config']['stage_ms_per_step']['policy'], d['config']['result_checksum']))"
  done
done

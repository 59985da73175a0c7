#!/bin/bash
# tools/diag/ab_cfg.sh <cfg: 2|3|5|3f> <variant>...   (cfg 3 = kuka7 2000 k10 x64 bf16, 5 = kuka14 5000 k16 x32 bf16, 3f = kuka7 fp32)
R=${GRAFT_REPO_ROOT:-$PWD}
case $1 in
 2) A="";;
 3) A="--env kuka7 --nodes 2000 --k1 10 --graphs 64 --mlp-dtype bf16";;
 5) A="--env kuka14 --nodes 5000 --k1 16 --graphs 32 --mlp-dtype bf16";;
 3f) A="--env kuka7 --nodes 2000 --k1 10 --graphs 64";;
esac
shift
for v in "$@"; do
  if [ "$v" = base ]; then unset GNNMP_LIB; else export GNNMP_LIB=$R/gnn-motion-planning_amd/libgnnmp_$v.so; fi
  python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --pcie-steps 0 --dense-steps 0 --bf16x3-steps 0 --single-steps 0 --inflight-steps 0 --other-configs-steps 0 --planner-problems 0 $A 2>&1 | tail -1 | python -c "
import json,sys
t=sys.stdin.read()
try:
    d=json.loads(t)
    print('cfg$1 %-10s %9.1f graphs/s  ms/step %.4f  stages %s  checksum %s' % ('$v', d['value'], d['ms_per_step'], d['config'].get('stage_ms_per_step'), d['config'].get('result_checksum')))
except Exception as e:
    print('cfg$1 $v FAILED:', t[-300:])"
done

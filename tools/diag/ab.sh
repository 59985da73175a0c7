#!/bin/bash
# A/B of experiment builds on the headline workload: tools/diag/ab.sh [bench args --] <variant> <variant> ...   ("base" = libgnnmp.so)
R=${GRAFT_REPO_ROOT:-$PWD}
ARGS=""
if [[ "$*" == *" -- "* ]]; then ARGS="${*%% -- *}"; set -- ${*#* -- }; fi
for v in "$@"; do
  if [ "$v" = base ]; then unset GNNMP_LIB; else export GNNMP_LIB=$R/gnn-motion-planning_amd/libgnnmp_$v.so; fi
  python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --pcie-steps 0 --dense-steps 0 --bf16x3-steps 0 --single-steps 0 --inflight-steps 0 --other-configs-steps 0 --planner-problems 0 --strong-leg 0 $ARGS 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-12s %9.1f graphs/s  ms/step %.4f  stages %s  checksum %s' % ('$v', d['value'], d['ms_per_step'], d['config'].get('stage_ms_per_step'), d['config'].get('result_checksum')))"
done

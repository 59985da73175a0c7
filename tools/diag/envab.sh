#!/bin/bash
# tools/diag/envab.sh <cfg: 2|3|5|3f> "<ENV=val ...>" ...   A/B of environment-variable switches on one BASELINE shape ("-" = none)
R=${GRAFT_REPO_ROOT:-$PWD}
case $1 in
 2) A="";;
 3) A="--env kuka7 --nodes 2000 --k1 10 --graphs 64 --mlp-dtype bf16";;
 5) A="--env kuka14 --nodes 5000 --k1 16 --graphs 32 --mlp-dtype bf16";;
 3f) A="--env kuka7 --nodes 2000 --k1 10 --graphs 64";;
esac
C=$1
shift
for v in "$@"; do
  E=""; [ "$v" != "-" ] && E="$v"
  env $E python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --pcie-steps 0 --dense-steps 0 --bf16x3-steps 0 --single-steps 0 --inflight-steps 0 --other-configs-steps 0 --planner-problems 0 --strong-leg 0 $A 2>&1 | tail -1 | python -c "
import json,sys
t=sys.stdin.read()
try:
    d=json.loads(t)
    print('cfg$C %-28s %9.1f graphs/s  ms/step %.4f  stages %s  checksum %s' % ('$v', d['value'], d['ms_per_step'], d['config'].get('stage_ms_per_step'), d['config'].get('result_checksum')))
except Exception as e:
    print('cfg$C $v FAILED:', t[-300:])"
done

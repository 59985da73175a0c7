#!/bin/bash
# Run ON THE GPU BOX: stage split of a kuka7 (d = 64) forward at 8 .. 32 problems of 1000 nodes, fp32 and bf16, with the message
# kernel's few-tiles form forced on / off -- where is the step between 8 and 16 problems of the cost sweep?
R=${GRAFT_REPO_ROOT:-$PWD}
BA="--steps 20 --warmup 5 --no-cpu-baseline --planner-problems 0 --strong-leg 0 --pcie-steps 0 --dense-steps 0 --bf16x3-steps 0 --single-steps 0 --inflight-steps 0 --other-configs-steps 0"
for dt in fp32 bf16; do for n in 8 12 16 20 24 32; do for coop in auto 0 1; do
  if [ $coop = auto ]; then unset GNNMP_MP_COOP; else export GNNMP_MP_COOP=$coop; fi
  python $R/bench.py $BA --env ${1:-kuka7} --nodes 1000 --k1 8 --graphs $n --mlp-dtype $dt 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$dt n=$n coop=$coop  %.4f ms  %s' % (d['ms_per_step'], d['config']['stage_ms_per_step']))"
done; done; done

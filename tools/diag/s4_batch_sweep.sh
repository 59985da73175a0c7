#!/bin/bash
# stage split of one forward against the batch size (the mixed job of BASELINE configs[3] runs 64-problem families; strong scaling shrinks
# the per-rank batch): where does the per-graph cost double between 256 and 64 graphs?
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/s4; mkdir -p $O
BA="--no-cpu-baseline --planner-problems 0 --strong-leg 0 --pcie-steps 0 --dense-steps 0 --bf16x3-steps 0 --single-steps 0 --inflight-steps 0 --other-configs-steps 0 --steps 20 --warmup 5"
{
for env in maze2 kuka7 ur5; do
for g in 16 32 64 128 256; do
  python $R/bench.py $BA --env $env --nodes 1000 --k1 8 --graphs $g 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('$env', $g, 'graphs/s %.0f ms %.4f' % (d['value'], d['ms_per_step']), ' '.join('%s %.4f' % kv for kv in c['stage_ms_per_step'].items()))"
done; done
} > $O/batch_sweep.txt 2>&1
cat $O/batch_sweep.txt

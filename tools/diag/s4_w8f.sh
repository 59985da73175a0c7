#!/bin/bash
# A/B of mp_fused_w8_kernel<64, 0> variants: bit-identity test + kuka7 fp32 bench lines
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/s4; mkdir -p $O
BA="--no-cpu-baseline --planner-problems 0 --strong-leg 0 --pcie-steps 0 --dense-steps 0 --bf16x3-steps 0 --single-steps 0 --inflight-steps 0 --other-configs-steps 0 --steps 20 --warmup 5"
{
true
for lib in "" $@; do unset GNNMP_LIB
  if [ -n "$lib" ]; then export GNNMP_LIB=$R/gnn-motion-planning_amd/$lib; fi
  for a in "--env kuka7 --nodes 2000 --k1 10 --graphs 64" "--env kuka7 --nodes 1000 --k1 8 --graphs 64"; do
  python $R/bench.py $BA $a 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('lib=$lib', '$a', 'graphs/s %.0f ms %.4f' % (d['value'], d['ms_per_step']), ' '.join('%s %.4f' % kv for kv in c['stage_ms_per_step'].items()), 'checksum', c.get('result_checksum'))"
  done
done
} > $O/w8f.txt 2>&1
cat $O/w8f.txt
{
for v in trace0 trace trace2; do
  echo "== $v"
  GNNMP_LIB=$R/gnn-motion-planning_amd/libgnnmp_$v.so python tools/diag/mp_trace2.py kuka7 2000 10 64 fp32 2>&1 | grep -v "GNNMP_LIB\|amdgpu.ids" | head -12
done
} > $O/w8f_trace.txt 2>&1
cat $O/w8f_trace.txt

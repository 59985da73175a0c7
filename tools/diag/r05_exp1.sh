#!/bin/bash
# round 5 experiment 1: node phase of mp_fused at d = 64 bf16 -- baseline timeline vs node weights in LDS (one WG per CU)
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p gpurun_out/r05
{
for v in trace trace64; do
  echo "=== $v"
  GNNMP_LIB=$R/gnn-motion-planning_amd/libgnnmp_$v.so python tools/diag/mp_trace.py kuka7 2000 10 64 bf16 2>&1 | grep -v amdgpu.ids
done
echo "=== A/B cfg3 shape"
bash tools/diag/ab.sh --env kuka7 --nodes 2000 --k1 10 --graphs 64 --mlp-dtype bf16 -- base lds64 base lds64
} > gpurun_out/r05/exp1.txt 2>&1
python - <<'PY' >> gpurun_out/r05/exp1.txt 2>&1
# single-graph latency after the status change
import subprocess, json, sys
out = subprocess.run([sys.executable, 'bench.py', '--steps', '5', '--warmup', '2', '--no-cpu-baseline', '--planner-problems', '0', '--strong-leg', '0', '--pcie-steps', '0',
                      '--dense-steps', '0', '--bf16x3-steps', '0'], capture_output=True, text=True).stdout
d = json.loads([l for l in out.splitlines() if l.startswith('{')][0])
print('single_graph_us', d['config']['single_graph_us'], 'value', d['value'])
PY
cat gpurun_out/r05/exp1.txt

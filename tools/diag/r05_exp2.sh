#!/bin/bash
mkdir -p gpurun_out/r05
{
python tools/diag/abx.py 3 base lds64
python tools/diag/abx.py 5 base
python tools/diag/abx.py 3f base
echo "=== single-graph latency (tools/diag/lat_single.py)"
python tools/diag/lat_single.py 2>&1 | grep -v amdgpu
echo "=== bench default, single-graph only"
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --planner-problems 0 --strong-leg 0 --pcie-steps 0 --dense-steps 0 --bf16x3-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(d['config']['single_graph_us'], d['value'])"
} > gpurun_out/r05/exp2.txt 2>&1
cat gpurun_out/r05/exp2.txt

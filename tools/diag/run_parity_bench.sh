python tools/parity_report.py fp32 > gpurun_out/parity_f64.txt 2>&1; sed -n 8,13p gpurun_out/parity_f64.txt; sed -n 16,17p gpurun_out/parity_f64.txt
python -m pytest tests/test_explorer_parity.py tests/test_explorer_fuzz_gpu.py -x -q 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_f64.txt 2>&1
python - <<PY
import json
d=json.loads(open("gpurun_out/bench_f64.txt").read().strip().splitlines()[-1])
print(d["value"], d["config"].get("stage_ms_per_step"), d["config"].get("single_graph_us"))
PY

set -u
mkdir -p gpurun_out/s4
BA="--no-cpu-baseline --planner-problems 0 --strong-leg 0 --pcie-steps 0 --dense-steps 0 --bf16x3-steps 0 --single-steps 0 --inflight-steps 0 --other-configs-steps 0"
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/s4/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/s4/pytest.log
tail -5 gpurun_out/s4/pytest.log
timeout 300 python bench.py $BA 2>/dev/null | tail -1 > gpurun_out/s4/b_cfg2.json
timeout 300 python bench.py $BA --env kuka7 --nodes 2000 --k1 10 --graphs 64 --mlp-dtype bf16 2>/dev/null | tail -1 > gpurun_out/s4/b_cfg3.json
timeout 300 python bench.py $BA --env kuka7 --nodes 2000 --k1 10 --graphs 64 2>/dev/null | tail -1 > gpurun_out/s4/b_cfg3f.json
timeout 300 python bench.py $BA --env kuka14 --nodes 5000 --k1 16 --graphs 32 --mlp-dtype bf16 2>/dev/null | tail -1 > gpurun_out/s4/b_cfg5.json
python tools/latency.py > gpurun_out/s4/latency.txt 2>&1
for f in gpurun_out/s4/b_*.json; do python - $f <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read()); print(sys.argv[1], d['value'], d['config']['stage_ms_per_step'])
P
done

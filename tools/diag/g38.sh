cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/g38.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> gpurun_out/g38.log 2>&1
timeout 900 python bench.py > gpurun_out/g38_bench.json 2> gpurun_out/g38_bench.err
tail -1 gpurun_out/g38_bench.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['planner']['device_planner']['timing'], d['cpu_baseline']['value'])" >> gpurun_out/g38.log

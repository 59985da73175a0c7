cd $GRAFT_REPO_ROOT
BA="--no-cpu-baseline --planner-problems 0 --pcie-steps 0 --dense-steps 0 --bf16x3-steps 0 --single-steps 0 --inflight-steps 0 --other-configs-steps 0"
GNNMP_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python bench.py $BA --strong 256 2>/dev/null | grep '^{' | tail -1 > gpurun_out/g19_strong.json
python tools/cfg5_pipeline.py 2>/dev/null | grep '^{' > gpurun_out/g19_cfg5pipe.json
timeout 600 python -m pytest tests/test_dist_gpu.py tests/test_cfg5_pipeline_gpu.py -x -q 2>&1 | tail -3
cat gpurun_out/g19_strong.json | cut -c1-900; cat gpurun_out/g19_cfg5pipe.json

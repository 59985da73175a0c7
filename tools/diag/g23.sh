cd $GRAFT_REPO_ROOT
bash tools/diag/ab_cfg.sh 3 base norl base norl 2>/dev/null > gpurun_out/g23.log
bash tools/diag/ab_cfg.sh 5 base norl base norl 2>/dev/null >> gpurun_out/g23.log
cat gpurun_out/g23.log
timeout 600 python -m pytest tests/test_explorer_bf16.py tests/test_full_size_bf16_gpu.py tests/test_full_size_gpu.py -x -q 2>&1 | tail -3

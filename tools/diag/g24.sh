cd $GRAFT_REPO_ROOT
bash tools/diag/ab_cfg.sh 3 base norl base 2>/dev/null > gpurun_out/g24.log
bash tools/diag/ab_cfg.sh 5 base norl base 2>/dev/null >> gpurun_out/g24.log
bash tools/diag/ab_cfg.sh 2 base norl base norl 2>/dev/null >> gpurun_out/g24.log
bash tools/diag/ab_cfg.sh 3f base norl 2>/dev/null >> gpurun_out/g24.log
cat gpurun_out/g24.log
timeout 900 python -m pytest tests/test_explorer_parity.py tests/test_explorer_bf16.py tests/test_explorer_bf16x3.py tests/test_full_size_bf16_gpu.py tests/test_full_size_gpu.py tests/test_full_size_mixed_gpu.py tests/test_explorer_fuzz_gpu.py -x -q 2>&1 | tail -3

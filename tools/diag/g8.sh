cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_explorer_parity.py tests/test_explorer_bf16.py tests/test_full_size_gpu.py tests/test_full_size_bf16_gpu.py tests/test_explorer_fuzz_gpu.py tests/test_hipgraph_gpu.py tests/test_explorer_bf16x3.py -x -q 2>&1 | tail -5 > gpurun_out/g8.log
bash tools/diag/ab_cfg.sh 2 base base old 2>/dev/null >> gpurun_out/g8.log
bash tools/diag/ab_cfg.sh 3 base old 2>/dev/null >> gpurun_out/g8.log
bash tools/diag/ab_cfg.sh 5 base old 2>/dev/null >> gpurun_out/g8.log
cat gpurun_out/g8.log

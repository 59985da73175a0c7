cd $GRAFT_REPO_ROOT
bash tools/diag/ab_cfg.sh 3 base old 2>/dev/null > gpurun_out/g6.log
bash tools/diag/ab_cfg.sh 5 base old 2>/dev/null >> gpurun_out/g6.log
bash tools/diag/ab_cfg.sh 2 base deep32 old 2>/dev/null >> gpurun_out/g6.log
cat gpurun_out/g6.log
timeout 900 python -m pytest tests/test_explorer_parity.py tests/test_explorer_bf16.py tests/test_full_size_gpu.py tests/test_full_size_bf16_gpu.py tests/test_explorer_fuzz_gpu.py tests/test_hipgraph_gpu.py -x -q 2>&1 | tail -5

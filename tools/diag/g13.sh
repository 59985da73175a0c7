cd $GRAFT_REPO_ROOT
bash tools/diag/ab_cfg.sh 2 base f32new old 2>/dev/null > gpurun_out/g13.log
GNNMP_MP_PAIR=1 bash tools/diag/ab_cfg.sh 2 base f32new 2>/dev/null >> gpurun_out/g13.log
bash tools/diag/envab.sh 3 - GNNMP_MP_PAIR=1 GNNMP_MP_COOP=2 GNNMP_MP_COOP=4 >> gpurun_out/g13.log 2>&1
bash tools/diag/envab.sh 5 - GNNMP_MP_COOP=2 GNNMP_MP_COOP=4 >> gpurun_out/g13.log 2>&1
cat gpurun_out/g13.log
timeout 900 python -m pytest tests/test_explorer_parity.py tests/test_explorer_bf16.py tests/test_full_size_gpu.py tests/test_explorer_fuzz_gpu.py -x -q 2>&1 | tail -3

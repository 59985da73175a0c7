cd $GRAFT_REPO_ROOT
bash tools/diag/envab.sh 3 - GNNMP_MP_ORDER=0 GNNMP_MP_ORDER=2 > gpurun_out/g14.log 2>&1
bash tools/diag/envab.sh 5 - GNNMP_MP_ORDER=0 GNNMP_MP_ORDER=1 >> gpurun_out/g14.log 2>&1
bash tools/diag/envab.sh 2 - GNNMP_MP_ORDER=0 GNNMP_MP_ORDER=2 >> gpurun_out/g14.log 2>&1
cat gpurun_out/g14.log
timeout 900 python -m pytest tests/test_full_size_gpu.py tests/test_explorer_parity.py tests/test_explorer_bf16.py tests/test_full_size_bf16_gpu.py -x -q 2>&1 | tail -3

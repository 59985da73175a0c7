cd $GRAFT_REPO_ROOT
bash tools/diag/ab_cfg.sh 3 base abl_NO_KE abl_NO_GATHER abl_NO_NODE abl_NO_ATOMICS abl_NO_EDGE 2>/dev/null > gpurun_out/g3.log
bash tools/diag/ab_cfg.sh 5 base abl_NO_KE abl_NO_GATHER abl_NO_NODE abl_NO_ATOMICS abl_NO_EDGE 2>/dev/null >> gpurun_out/g3.log
cat gpurun_out/g3.log
timeout 1200 python -m pytest tests/test_parity_census_gpu.py tests/test_cfg5_pipeline_gpu.py tests/test_full_size_mixed_gpu.py tests/test_explorer_autograd_gpu.py -x -q -s 2>&1 | tail -80 > gpurun_out/g3_tests.log
tail -5 gpurun_out/g3_tests.log
python tools/mixed_bench.py > gpurun_out/g3_mixed.txt 2>&1
python tools/cfg5_pipeline.py > gpurun_out/g3_cfg5pipe.txt 2>&1
cat gpurun_out/g3_mixed.txt gpurun_out/g3_cfg5pipe.txt

cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/g16.log
bash tools/diag/envab.sh 2 - >> gpurun_out/g16.log 2>&1
bash tools/diag/envab.sh 3 - >> gpurun_out/g16.log 2>&1
bash tools/diag/envab.sh 5 - >> gpurun_out/g16.log 2>&1
python tools/parity_census.py > gpurun_out/g16_census.txt 2>&1
python tools/cfg5_pipeline.py > gpurun_out/g16_cfg5pipe.txt 2>&1
cat gpurun_out/g16.log; tail -3 gpurun_out/g16_census.txt; cat gpurun_out/g16_cfg5pipe.txt

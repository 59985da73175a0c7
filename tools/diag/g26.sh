#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/g26.log
bash tools/diag/envab.sh 3 - - >> gpurun_out/g26.log 2>&1
bash tools/diag/envab.sh 5 - >> gpurun_out/g26.log 2>&1
bash tools/diag/envab.sh 2 - >> gpurun_out/g26.log 2>&1
python tools/mixed_bench.py 2>&1 | tail -12 >> gpurun_out/g26.log

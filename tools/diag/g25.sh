#!/bin/bash
# order-2 (persistent snake) A/B on the three shapes + ragged-order test
cd $GRAFT_REPO_ROOT
bash tools/diag/envab.sh 3 - GNNMP_MP_ORDER=2 GNNMP_MP_ORDER=1 GNNMP_MP_ORDER=0 GNNMP_MP_ORDER=2 > gpurun_out/g25.log 2>&1
bash tools/diag/envab.sh 5 - GNNMP_MP_ORDER=2 GNNMP_MP_ORDER=0 >> gpurun_out/g25.log 2>&1
bash tools/diag/envab.sh 2 - GNNMP_MP_ORDER=2 >> gpurun_out/g25.log 2>&1
bash tools/diag/envab.sh 3f - GNNMP_MP_ORDER=2 >> gpurun_out/g25.log 2>&1
GNNMP_MP_ORDER=2 timeout 600 python -m pytest tests/test_full_size_gpu.py tests/test_explorer_gpu.py -m gpu -x -q 2>&1 | tail -5 >> gpurun_out/g25.log

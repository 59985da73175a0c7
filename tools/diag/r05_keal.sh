#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05; mkdir -p $O
{
python tools/diag/abx.py 3 base ntw keal kealw
python tools/diag/abx.py 3f base ntw keal kealw
} > $O/keal.txt 2>&1
cat $O/keal.txt
bash tools/diag/r05_sizes2.sh keal kealw | grep -A3 "^=="

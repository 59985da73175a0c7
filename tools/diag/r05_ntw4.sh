#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05; mkdir -p $O
{
python tools/diag/abx.py 5 base ntw4
python tools/diag/abx.py 2 base ntw4
python tools/diag/abx.py 3 base
python tools/diag/abx.py 2b base ntw4
} > $O/ntw4.txt 2>&1
cat $O/ntw4.txt

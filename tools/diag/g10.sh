cd $GRAFT_REPO_ROOT
bash tools/diag/envab.sh 3 - GNNMP_MP_PAIR=1 - GNNMP_MP_PAIR=1 > gpurun_out/g10.log 2>&1
bash tools/diag/envab.sh 5 - GNNMP_MP_PAIR=1 >> gpurun_out/g10.log 2>&1
bash tools/diag/envab.sh 2 - GNNMP_MP_PAIR=1 >> gpurun_out/g10.log 2>&1
cat gpurun_out/g10.log

cd $GRAFT_REPO_ROOT
bash tools/diag/envab.sh 3 - GNNMP_MP_GPG=16 GNNMP_MP_GPG=8 - > gpurun_out/g2.log 2>&1
bash tools/diag/envab.sh 5 - GNNMP_MP_GPG=40 GNNMP_MP_GPG=20 GNNMP_MP_GPG=8 >> gpurun_out/g2.log 2>&1
bash tools/diag/envab.sh 2 - GNNMP_MP_GPG=8 GNNMP_MP_GPG=4 >> gpurun_out/g2.log 2>&1
cat gpurun_out/g2.log

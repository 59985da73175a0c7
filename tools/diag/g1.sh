cd $GRAFT_REPO_ROOT
bash tools/diag/envab.sh 2 - > gpurun_out/g1.log 2>&1
bash tools/diag/envab.sh 3 - GNNMP_MP_COOP=2 GNNMP_MP_COOP=4 - >> gpurun_out/g1.log 2>&1
bash tools/diag/envab.sh 5 - GNNMP_MP_COOP=2 GNNMP_MP_COOP=4 - >> gpurun_out/g1.log 2>&1
bash tools/diag/pmc_mp.sh > gpurun_out/g1_pmc.log 2>&1
cp gpurun_out/pmc_mp_cfg3/summary.txt gpurun_out/g1_pmc_summary.txt
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 >> gpurun_out/g1.log
cat gpurun_out/g1.log

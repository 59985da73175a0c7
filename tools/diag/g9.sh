cd $GRAFT_REPO_ROOT
bash tools/diag/pmc_mp.sh > gpurun_out/g9_pmc.log 2>&1
cp gpurun_out/pmc_mp_cfg3/summary.txt gpurun_out/g9_pmc_summary.txt
grep -A22 "mp_fused_kernel<64, 1, 1>" gpurun_out/g9_pmc_summary.txt | head -24

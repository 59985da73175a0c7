cd $GRAFT_REPO_ROOT
GNNMP_MP_ORDER=0 bash tools/diag/ab_cfg.sh 2 base nl old base nl old 2>/dev/null > gpurun_out/g15.log
bash tools/diag/ab_cfg.sh 2 base 2>/dev/null >> gpurun_out/g15.log
cat gpurun_out/g15.log

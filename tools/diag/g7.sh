cd $GRAFT_REPO_ROOT
for v in base noae nol l2 deep32; do
  if [ "$v" = base ]; then unset GNNMP_LIB; else export GNNMP_LIB=$GRAFT_REPO_ROOT/gnn-motion-planning_amd/libgnnmp_$v.so; fi
  echo "== $v"
  timeout 300 python -m pytest tests/test_explorer_parity.py -x -q -k "golden_scores" 2>&1 | tail -3
done > gpurun_out/g7.log 2>&1
unset GNNMP_LIB
bash tools/diag/ab_cfg.sh 2 base noae nol l2 2>/dev/null >> gpurun_out/g7.log
cat gpurun_out/g7.log

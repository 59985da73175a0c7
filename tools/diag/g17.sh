cd $GRAFT_REPO_ROOT
bash tools/diag/ab_cfg.sh 3 base pl base pl 2>/dev/null > gpurun_out/g17.log
bash tools/diag/ab_cfg.sh 5 base pl 2>/dev/null >> gpurun_out/g17.log
bash tools/diag/ab_cfg.sh 2 base pl 2>/dev/null >> gpurun_out/g17.log
GNNMP_MP_ORDER=0 GNNMP_LIB=$GRAFT_REPO_ROOT/gnn-motion-planning_amd/libgnnmp_trace.so python tools/diag/mp_trace.py kuka7 2000 10 64 bf16 2>&1 | grep -v "GNNMP_LIB\|amdgpu" >> gpurun_out/g17.log
cat gpurun_out/g17.log
( time timeout 900 python -m pytest tests/test_parity_census_gpu.py -x -q 2>&1 | tail -3 ) 2>&1 | tail -8

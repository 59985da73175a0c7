cd $GRAFT_REPO_ROOT
bash tools/diag/ab_cfg.sh 3 base nowpre base nowpre 2>/dev/null > gpurun_out/g12.log
GNNMP_MP_PAIR=1 bash tools/diag/ab_cfg.sh 3 base 2>/dev/null >> gpurun_out/g12.log
export GNNMP_LIB=$GRAFT_REPO_ROOT/gnn-motion-planning_amd/libgnnmp_trace.so
python tools/diag/mp_trace.py kuka7 2000 10 64 bf16 2>&1 | grep -v "GNNMP_LIB\|amdgpu" >> gpurun_out/g12.log
unset GNNMP_LIB
cat gpurun_out/g12.log
timeout 600 python -m pytest tests/test_explorer_bf16.py tests/test_full_size_bf16_gpu.py -x -q 2>&1 | tail -3

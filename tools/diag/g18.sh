cd $GRAFT_REPO_ROOT
export GNNMP_LIB=$GRAFT_REPO_ROOT/gnn-motion-planning_amd/libgnnmp_trace.so
GNNMP_MP_ORDER=0 python tools/diag/mp_trace.py kuka7 2000 10 64 bf16 2>&1 | grep -v "GNNMP_LIB\|amdgpu" > gpurun_out/g18.log
python tools/diag/mp_trace.py kuka7 2000 10 64 bf16 2>&1 | grep -v "GNNMP_LIB\|amdgpu" >> gpurun_out/g18.log
python tools/diag/mp_trace.py kuka14 5000 16 32 bf16 2>&1 | grep -v "GNNMP_LIB\|amdgpu" >> gpurun_out/g18.log
cat gpurun_out/g18.log

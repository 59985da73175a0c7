cd $GRAFT_REPO_ROOT
export GNNMP_LIB=$GRAFT_REPO_ROOT/gnn-motion-planning_amd/libgnnmp_trace.so
python tools/diag/mp_trace.py kuka7 2000 10 64 bf16 > gpurun_out/g11.log 2>&1
GNNMP_MP_PAIR=1 python tools/diag/mp_trace.py kuka7 2000 10 64 bf16 >> gpurun_out/g11.log 2>&1
python tools/diag/mp_trace.py maze2 1000 8 256 fp32 >> gpurun_out/g11.log 2>&1
python tools/diag/mp_trace.py kuka14 5000 16 32 bf16 >> gpurun_out/g11.log 2>&1
grep -v "GNNMP_LIB\|amdgpu.ids" gpurun_out/g11.log

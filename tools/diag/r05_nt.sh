#!/bin/bash
# non-temporal node-row loads / stores in mp_fused_w8 (does the gathered A table stay in L2?): rate + L2 counters per variant
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05; mkdir -p $O
{
python tools/diag/abx.py 3 base ntr ntw ntrw
python tools/diag/abx.py 3f base ntrw
BA="--steps 3 --warmup 1 --no-cpu-baseline --planner-problems 0 --strong-leg 0 --pcie-steps 0 --dense-steps 0 --bf16x3-steps 0 --single-steps 0 --inflight-steps 0 --other-configs-steps 0 --env kuka7 --nodes 2000 --k1 10 --graphs 64 --mlp-dtype bf16"
for v in base ntrw; do
  if [ $v != base ]; then export GNNMP_LIB=$R/gnn-motion-planning_amd/libgnnmp_$v.so; fi
  PMC_GROUPS="FETCH_SIZE;WRITE_SIZE;TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" bash tools/pmc_passes.sh nt_$v -- python $R/bench.py $BA
  echo "== $v"; grep -A8 "mp_fused_w8_kernel<64, 1" $R/gpurun_out/pmc_nt_$v/summary.txt
done
} > $O/nt.txt 2>&1
cat $O/nt.txt

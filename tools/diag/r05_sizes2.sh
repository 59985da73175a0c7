#!/bin/bash
# traffic attribution of mp_fused_w8 (configs[2] shape): 128-byte memory-side reads with the A gather replaced by the tile's own rows
# (aown), with temporal K_e loads (ket), with non-temporal row stores (ntw)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05; mkdir -p $O
G="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_HIT_sum TCC_MISS_sum"
BA="--steps 3 --warmup 1 --no-cpu-baseline --planner-problems 0 --strong-leg 0 --pcie-steps 0 --dense-steps 0 --bf16x3-steps 0 --single-steps 0 --inflight-steps 0 --other-configs-steps 0 --env kuka7 --nodes 2000 --k1 10 --graphs 64 --mlp-dtype bf16"
{
for v in "$@"; do
  unset GNNMP_LIB
  if [ $v != base ]; then export GNNMP_LIB=$R/gnn-motion-planning_amd/libgnnmp_$v.so; fi
  PMC_GROUPS="$G" bash tools/pmc_passes.sh sz_$v -- python $R/bench.py $BA
  echo "== $v"; grep -A5 "mp_fused_w8_kernel<64, 1" $R/gpurun_out/pmc_sz_$v/summary.txt
done
} > $O/sizes2.txt 2>&1
cat $O/sizes2.txt

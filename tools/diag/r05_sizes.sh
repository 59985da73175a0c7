#!/bin/bash
# size-resolved memory-side read requests (32 / 64 / 128 B) of mp_fused_w8 at the configs[2] shape, with a calibration on known byte counts
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05; mkdir -p $O
G="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum;TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum;FETCH_SIZE;WRITE_SIZE"
{
PMC_GROUPS="$G" bash tools/pmc_passes.sh calib -- python $R/tools/diag/traffic_calib.py
cat $R/gpurun_out/pmc_calib/summary.txt | grep -v "^$" | head -60
BA="--steps 3 --warmup 1 --no-cpu-baseline --planner-problems 0 --strong-leg 0 --pcie-steps 0 --dense-steps 0 --bf16x3-steps 0 --single-steps 0 --inflight-steps 0 --other-configs-steps 0 --env kuka7 --nodes 2000 --k1 10 --graphs 64 --mlp-dtype bf16"
for v in base ntw; do
  if [ $v != base ]; then export GNNMP_LIB=$R/gnn-motion-planning_amd/libgnnmp_$v.so; fi
  PMC_GROUPS="$G" bash tools/pmc_passes.sh sz_$v -- python $R/bench.py $BA
  echo "== $v"; grep -A9 "mp_fused_w8_kernel<64, 1" $R/gpurun_out/pmc_sz_$v/summary.txt
done
} > $O/sizes.txt 2>&1
cat $O/sizes.txt

#!/bin/bash
# Run ON THE GPU BOX: SQ / traffic counters of mp_fused at the configs[2] shape in bf16 (kuka7 2000-node k=10 x 64)
R=${GRAFT_REPO_ROOT:-$PWD}
export PMC_GROUPS="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES;SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR;FETCH_SIZE;WRITE_SIZE;GRBM_GUI_ACTIVE"
bash tools/pmc_passes.sh mp_cfg3 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --planner-problems 0 --pcie-steps 0 --dense-steps 0 --bf16x3-steps 0 --single-steps 0 --inflight-steps 0 --other-configs-steps 0 --env kuka7 --nodes 2000 --k1 10 --graphs 64 --mlp-dtype bf16
grep -A22 "mp_fused_kernel<64, 1, 1>" gpurun_out/pmc_mp_cfg3/summary.txt | head -24

#!/bin/bash
# PMC counters of the obstacle launch (obs role + node_f64 role) on the headline workload
export PMC_GROUPS="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES;SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU;SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_LDS;GRBM_GUI_ACTIVE"
bash tools/pmc_passes.sh obs -- python $PWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --pcie-steps 0 --dense-steps 0 --bf16x3-steps 0 --single-steps 0 --inflight-steps 0 --other-configs-steps 0 --planner-problems 0
grep -A40 "obs_kernel" gpurun_out/pmc_obs/summary.txt | head -60

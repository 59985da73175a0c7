#!/bin/bash
# Run ON THE GPU BOX: counters + launch times of the smoother's batched fp32 forward (256 problems, C = 14) with the streamed-weights
# message kernel on (default) / off
R=${GRAFT_REPO_ROOT:-$PWD}
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"
G2="SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
for s in ${@:-1 0}; do
  export GNNMP_SM_STREAM=$s
  PMC_GROUPS="$G1;$G2;GRBM_GUI_ACTIVE" bash tools/pmc_passes.sh smst$s -- python $R/tools/diag/smoother_c14.py 14fp32 > gpurun_out/pmc_smst$s.log 2>&1
  echo "== GNNMP_SM_STREAM=$s"; grep -A20 "sm_msg" gpurun_out/pmc_smst$s/summary.txt | head -44
done

#!/bin/bash
# Run ON THE GPU BOX: LDS / issue counters of the two bf16 shapes (and cfg 2) -> gpurun_out/pmc_lds_<tag>.txt
R=${GRAFT_REPO_ROOT:-$PWD}
BA="--steps 3 --warmup 1 --no-cpu-baseline --planner-problems 0 --strong-leg 0 --pcie-steps 0 --dense-steps 0 --bf16x3-steps 0 --single-steps 0 --inflight-steps 0 --other-configs-steps 0"
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"
G2="SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
G3="SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"
for t in "${@:-c3 c5}"; do
  case $t in
    c2) A="";;
    c3) A="--env kuka7 --nodes 2000 --k1 10 --graphs 64 --mlp-dtype bf16";;
    c5) A="--env kuka14 --nodes 5000 --k1 16 --graphs 32 --mlp-dtype bf16";;
    c3f) A="--env kuka7 --nodes 2000 --k1 10 --graphs 64";;
  esac
  PMC_GROUPS="$G1;$G2;$G3" bash tools/pmc_passes.sh lds_$t -- python $R/bench.py $BA $A > gpurun_out/pmc_lds_$t.log 2>&1
  cp gpurun_out/pmc_lds_$t/summary.txt gpurun_out/pmc_lds_$t.txt
  python - <<P
import re
cur=None; d={}
for ln in open('gpurun_out/pmc_lds_$t.txt'):
    if ln and not ln.startswith(' '): cur=ln.strip(); continue
    m=re.match(r'\s+(\S+)\s+per-dispatch\s+([0-9.eE+-]+)',ln)
    if m and cur: d.setdefault(cur,{})[m.group(1)]=float(m.group(2))
for k,v in d.items():
    if 'mp_fused' in k or 'policy' in k or 'pre_resident' in k or 'pre_kernel' in k:
        a=v.get('SQ_ACTIVE_INST_LDS',0); c=v.get('SQ_LDS_BANK_CONFLICT',0)
        print('$t %-60s conflict %.3g  lds_active %.3g  ratio %.3f  idx_active %.3g  insts_lds %.3g' % (k[:60], c, a, c/max(a,1), v.get('SQ_LDS_IDX_ACTIVE',0), v.get('SQ_INSTS_LDS',0)))
P
done

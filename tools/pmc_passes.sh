#!/bin/bash
# Run ON THE GPU BOX: several rocprofv3 --pmc passes (each within the per-block slot limits of gfx950) of one command,
# then a per-kernel table.   tools/pmc_passes.sh <tag> -- <command...>
# Counter groups are in $PMC_GROUPS (';'-separated) or the default set below.  --kernel-trace only, as gpurun requires.
set -u
TAG=$1; shift; shift
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
GROUPS_DEFAULT="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES;SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU;FETCH_SIZE;WRITE_SIZE;TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum;TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum;TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum;GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum"
G=${PMC_GROUPS:-$GROUPS_DEFAULT}
cd /tmp && export TMPDIR=/tmp
i=0
IFS=';' read -ra ARR <<< "$G"
for grp in "${ARR[@]}"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $grp -d $OUT/p$i -o p -- "$@" > $OUT/p$i.log 2>&1 || echo "pass $i ($grp) failed: $(tail -2 $OUT/p$i.log)"
done
cd $R
python tools/pmc_summary.py $(find $OUT -name "p_results.db") -o $OUT/summary.txt > /dev/null
find $OUT -name "*.db" -delete      # the .db files are large; the summary and logs are kept
wc -l $OUT/summary.txt

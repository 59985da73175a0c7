#!/bin/bash
# Run ON THE GPU BOX: the MFMA-operation counter passes of the three bench shapes -> profiles/kernel_mfma.json (+ summaries under gpurun_out/mfma/)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/mfma; mkdir -p $O
BA="--steps 3 --warmup 1 --no-cpu-baseline --planner-problems 0 --strong-leg 0 --pcie-steps 0 --dense-steps 0 --bf16x3-steps 0 --single-steps 0 --inflight-steps 0 --other-configs-steps 0"
G="SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_MOPS_F16"
rocprofv3 -L 2>/dev/null | grep -o "SQ_INSTS_VALU_MFMA[A-Z0-9_]*\|SQ_VALU_MFMA[A-Z0-9_]*" | sort -u > $O/avail.txt
for t in "cfg2|maze2 N=1000 k1=8 graphs=256 fp32|" "cfg3|kuka7 N=2000 k1=10 graphs=64 bf16|--env kuka7 --nodes 2000 --k1 10 --graphs 64 --mlp-dtype bf16" \
         "cfg3f|kuka7 N=2000 k1=10 graphs=64 fp32|--env kuka7 --nodes 2000 --k1 10 --graphs 64" "cfg5|kuka14 N=5000 k1=16 graphs=32 bf16|--env kuka14 --nodes 5000 --k1 16 --graphs 32 --mlp-dtype bf16"; do
  IFS='|' read -r tag wk extra <<< "$t"
  PMC_GROUPS="$G" bash tools/pmc_passes.sh mfma_$tag -- python $R/bench.py $BA $extra > $O/$tag.log 2>&1
  cp gpurun_out/pmc_mfma_$tag/summary.txt $O/pmc_mfma_$tag.txt
  python tools/mfma_json.py $O/pmc_mfma_$tag.txt --workload "$wk" --steps-in-run 8 > $O/mfma_$tag.txt 2>&1
done
cp profiles/kernel_mfma.json $O/

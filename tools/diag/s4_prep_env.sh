#!/bin/bash
# big-graph prep: column-split hist + scatter (default) against the target-sliced one-launch kernel (GNNMP_PREP_PARTS=1, GNNMP_PREP_SLICES=n)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/s4; mkdir -p $O
{
for cfg in 5 3; do
  timeout 250 python tools/diag/abx.py $cfg base base,GNNMP_PREP_PARTS=1,GNNMP_PREP_SLICES=4 base,GNNMP_PREP_PARTS=1,GNNMP_PREP_SLICES=8 base,GNNMP_PREP_PARTS=1,GNNMP_PREP_SLICES=15 base,GNNMP_PREP_PARTS=4 base,GNNMP_PREP_PARTS=16
done
} > $O/prep_env.txt 2>&1
cut -c1-260 $O/prep_env.txt

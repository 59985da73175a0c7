#!/bin/bash
# per-dispatch facts of mp_fused (grid, LDS, registers, duration) for the base and lds64 builds at the configs[2] shape
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp; export TMPDIR=/tmp
for v in ${VARIANTS:-base lds64}; do
  if [ $v = base ]; then unset GNNMP_LIB; else export GNNMP_LIB=$R/gnn-motion-planning_amd/libgnnmp_$v.so; fi
  rm -rf /tmp/prof_$v
  rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$v -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --pcie-steps 0 --dense-steps 0 --bf16x3-steps 0 --single-steps 0 --inflight-steps 0 --other-configs-steps 0 --planner-problems 0 --strong-leg 0 --env kuka7 --nodes 2000 --k1 10 --graphs 64 --mlp-dtype bf16 > /dev/null 2>&1
  f=$(find /tmp/prof_$v -name "*kernel_trace.csv" | head -1)
  echo "== $v ($f)"
  python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
mp = [r for r in rows if 'mp_fused' in r['Kernel_Name']]
print(list(mp[0].keys()))
for r in mp[-6:]:
    print(r['Kernel_Name'][:40], 'grid', r.get('Grid_Size'), 'wg', r.get('Workgroup_Size'), 'lds', r.get('LDS_Block_Size'), 'vgpr', r.get('VGPR_Count'), 'accum', r.get('Accum_VGPR_Count'), 'scratch', r.get('Scratch_Size'),
          'us %.1f' % ((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
PY
done

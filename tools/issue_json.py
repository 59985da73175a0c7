#!/usr/bin/env python
"""profiles/kernel_issue.json from a tools/pmc_passes.sh summary: issue-slot accounting of one kernel on one workload.

    python tools/issue_json.py <summary.txt> --kernel-like 'pre_resident_kernel<64, 1, true' --stage edge_pre \\
        --workload 'kuka7 N=2000 k1=10 graphs=64 bf16' --tiles <32-row tiles per launch>

On this chip the matrix pipe and the other VALU instructions of a SIMD's waves ADD UP, so for a kernel bound by
instruction issue the honest roof is the SIMD's issue time: issue_slot_frac = (4 x SQ_ACTIVE_INST_VALU [quad-cycles] +
SQ_VALU_MFMA_BUSY_CYCLES [cycles]) / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs).  Entries are stamped like profiles/kernel_traffic.json."""
import argparse, datetime, json, os, re, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bench import kernel_source_hash  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('summary')
    ap.add_argument('--kernel-like', required=True)
    ap.add_argument('--stage', default='edge_pre')
    ap.add_argument('--workload', required=True)
    ap.add_argument('--tiles', type=float, default=0.0)
    a = ap.parse_args()
    vals, cur = {}, None
    for ln in open(a.summary):
        if ln and not ln.startswith(' '):
            cur = ln.strip()
            continue
        m = re.match(r'\s+(\S+)\s+per-dispatch\s+([0-9.eE+-]+)', ln)
        if m and cur and a.kernel_like in cur:
            vals[m.group(1)] = float(m.group(2))
    need = ('SQ_ACTIVE_INST_VALU', 'SQ_VALU_MFMA_BUSY_CYCLES', 'GRBM_GUI_ACTIVE', 'SQ_INSTS_VALU', 'SQ_INSTS_MFMA')
    if any(k not in vals for k in need):
        raise SystemExit('missing counters for %r: have %s' % (a.kernel_like, sorted(vals)))
    simd_cycles = vals['GRBM_GUI_ACTIVE'] / 8.0 * 1024.0
    valu = 4.0 * vals['SQ_ACTIVE_INST_VALU'] / simd_cycles
    mfma = vals['SQ_VALU_MFMA_BUSY_CYCLES'] / simd_cycles
    e = {'kernel_like': a.kernel_like, 'stage': a.stage, 'workload': a.workload, 'measured': datetime.date.today().isoformat(),
         'kernel_source_sha256': kernel_source_hash(), 'issue_slot_frac': round(valu + mfma, 4), 'valu_frac': round(valu, 4),
         'mfma_frac': round(mfma, 4), 'valu_insts_per_launch': vals['SQ_INSTS_VALU'], 'mfma_insts_per_launch': vals['SQ_INSTS_MFMA'],
         'valu_per_32_row_tile': round(vals['SQ_INSTS_VALU'] / a.tiles, 1) if a.tiles else None,
         'mfma_per_32_row_tile': round(vals['SQ_INSTS_MFMA'] / a.tiles, 1) if a.tiles else None,
         'source': 'rocprofv3 --kernel-trace --pmc passes (tools/pmc_passes.sh), per-dispatch averages'}
    path = os.path.join(REPO, 'profiles', 'kernel_issue.json')
    old = json.load(open(path)) if os.path.exists(path) else []
    old = [o for o in old if not (o.get('kernel_like') == a.kernel_like and o.get('workload') == a.workload)]
    old.append(e)
    json.dump(old, open(path, 'w'), indent=1)
    print(json.dumps(e))


if __name__ == '__main__':
    main()

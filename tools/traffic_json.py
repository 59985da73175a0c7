#!/usr/bin/env python
"""profiles/kernel_traffic.json from FETCH_SIZE / WRITE_SIZE passes (tools/profile_round.sh, tools/diag/traffic_pass.sh):
    python tools/traffic_json.py --kernel-like 'pre_resident_kernel<32, 0, true' --workload 'maze2 N=1000 k1=8 graphs=256 fp32' \\
        gpurun_out/pmc_fetch/**/p_results.db gpurun_out/pmc_write/**/p_results.db
HBM bytes per launch of one kernel on one workload, corrected as /opt/skills/guides/MI355X_MICROARCH.md (HBM section)
prescribes: counters in KiB, FETCH_SIZE doubled on gfx950 for wide coalesced reads, WRITE_SIZE as is.  Every entry is
stamped with the date of the pass and the hash of the kernel sources so bench.py can label the number and tell when it
went stale; an entry for the same (kernel, workload) is replaced."""
import argparse
import datetime
import json
import os
import sqlite3
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bench import kernel_source_hash  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--kernel-like', default='pre_resident_kernel<32, 0, true')
    ap.add_argument('--workload', default='maze2 N=1000 k1=8 graphs=256 fp32')
    ap.add_argument('--note', default='')
    ap.add_argument('dbs', nargs='+')
    a = ap.parse_args()
    vals, disp = {}, {}
    for path in a.dbs:
        cur = sqlite3.connect(path).cursor()
        q = "select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name"
        for k, c, v, n in cur.execute(q):
            if a.kernel_like in k and c in ('FETCH_SIZE', 'WRITE_SIZE'):
                vals[c] = v / n
                disp[c] = n
    if len(vals) != 2:
        raise SystemExit('need both FETCH_SIZE and WRITE_SIZE for %r, got %s' % (a.kernel_like, vals))
    entry = {
        'kernel_like': a.kernel_like, 'workload': a.workload,
        'source': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes); ' + a.note,
        'FETCH_SIZE_per_dispatch_KB': round(vals['FETCH_SIZE'], 1), 'WRITE_SIZE_per_dispatch_KB': round(vals['WRITE_SIZE'], 1),
        'dispatches': disp,
        'correction': 'counters are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of a wide (16 B/lane) coalesced '
                      'read, so the read side is doubled; WRITE_SIZE is taken as is',
        'hbm_bytes_per_launch': int(round((2 * vals['FETCH_SIZE'] + vals['WRITE_SIZE']) * 1024)),
        'measured': datetime.date.today().isoformat(),
        'kernel_source_sha256': kernel_source_hash(),
    }
    path = os.path.join(REPO, 'profiles', 'kernel_traffic.json')
    entries = json.load(open(path)) if os.path.exists(path) else []
    entries = [e for e in entries if (e.get('kernel_like'), e.get('workload')) != (a.kernel_like, a.workload)] + [entry]
    json.dump(entries, open(path, 'w'), indent=1)
    print(json.dumps(entry, indent=1))


if __name__ == '__main__':
    main()

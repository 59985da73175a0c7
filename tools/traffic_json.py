#!/usr/bin/env python
"""profiles/edge_pre_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of tools/profile_round.sh:
    python tools/traffic_json.py gpurun_out/pmc_fetch/**/p_results.db gpurun_out/pmc_write/**/p_results.db
HBM bytes per launch of the dominant kernel, corrected as /opt/skills/guides/MI355X_MICROARCH.md (HBM section)
prescribes: counters in KiB, FETCH_SIZE doubled on gfx950 for wide coalesced reads, WRITE_SIZE as is.  The file is
stamped with the hash of the kernel sources so bench.py can tell when it went stale."""
import json
import os
import sqlite3
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bench import kernel_source_hash  # noqa: E402


def main(paths, kernel_like='pre_resident_kernel<32, 0, true'):
    vals = {}
    for path in paths:
        cur = sqlite3.connect(path).cursor()
        q = "select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name"
        for k, c, v, n in cur.execute(q):
            if kernel_like in k and c in ('FETCH_SIZE', 'WRITE_SIZE'):
                vals[c] = v / n
    if len(vals) != 2:
        raise SystemExit('need both FETCH_SIZE and WRITE_SIZE for %r, got %s' % (kernel_like, vals))
    out = {
        'kernel': kernel_like + '...> (edge encoders + 3 obstacle-attention blocks)',
        'source': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes); bench.py --steps 3 --warmup 1 --unique 64',
        'FETCH_SIZE_per_dispatch_KB': round(vals['FETCH_SIZE'], 1), 'WRITE_SIZE_per_dispatch_KB': round(vals['WRITE_SIZE'], 1),
        'correction': 'counters are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of a wide (16 B/lane) coalesced '
                      'read, so the read side is doubled; WRITE_SIZE is taken as is',
        'hbm_bytes_per_launch': int(round((2 * vals['FETCH_SIZE'] + vals['WRITE_SIZE']) * 1024)),
        'kernel_source_sha256': kernel_source_hash(),
    }
    json.dump(out, open(os.path.join(REPO, 'profiles', 'edge_pre_traffic.json'), 'w'), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main(sys.argv[1:])

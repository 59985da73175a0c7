#!/usr/bin/env python
"""Summarise a rocprofv3 results .db (rocpd sqlite) into a per-kernel stats table
(calls, total/avg/min/max duration) -- the same content as `--stats` CSV output."""
import sqlite3
import sys


def main(path, out=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = 'name' if 'name' in cols else [c for c in cols if 'name' in c][0]
    rows = cur.execute("select %s, start, end from kernels" % name_col).fetchall()
    agg = {}
    for n, s, e in rows:
        a = agg.setdefault(n, [0, 0, 1 << 62, 0])
        d = e - s
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values()) or 1
    lines = ['%-78s %8s %14s %12s %12s %12s %7s' % ('kernel', 'calls', 'total_ns', 'avg_ns', 'min_ns', 'max_ns', 'pct')]
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        short = n if len(n) <= 78 else n[:75] + '...'
        lines.append('%-78s %8d %14d %12.0f %12d %12d %6.2f%%' % (short, a[0], a[1], a[1] / a[0], a[2], a[3], 100.0 * a[1] / tot))
    text = '\n'.join(lines)
    print(text)
    if out:
        open(out, 'w').write(text + '\n')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)

#!/usr/bin/env python
"""Per-kernel averages of the PMC counters in one or more rocprofv3 results .db files."""
import collections
import sqlite3
import sys


def main(paths, out=None):
    agg = collections.defaultdict(dict)
    for path in paths:
        cur = sqlite3.connect(path).cursor()
        q = "select kernel_name, counter_name, sum(value), count(*), avg(end - start) from counters_collection " \
            "group by kernel_name, counter_name"
        for k, c, v, n, d in cur.execute(q):
            agg[k][c] = (v / n, n, d)
    lines = []
    for k in sorted(agg, key=lambda kk: -max(x[0] for x in agg[kk].values())):
        if 'gnnmp' not in k:
            continue
        lines.append(k if len(k) < 110 else k[:107] + '...')
        for c, (v, n, d) in sorted(agg[k].items()):
            lines.append('    %-28s per-dispatch %18.1f   (dispatches %d, avg %.1f us)' % (c, v, n, d / 1000.0))
    text = '\n'.join(lines)
    print(text)
    if out:
        open(out, 'w').write(text + '\n')


if __name__ == '__main__':
    args = sys.argv[1:]
    out = None
    if '-o' in args:
        i = args.index('-o')
        out = args[i + 1]
        args = args[:i] + args[i + 2:]
    main(args, out)

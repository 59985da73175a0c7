#!/usr/bin/env python
"""Summarise a rocprofv3 results .db (rocpd sqlite) into a per-kernel stats table
(calls, total/avg/min/max duration) -- the same content as `--stats` CSV output.

    python tools/rocprof_summary.py <results.db> [out.txt] [--warmup W --steps K]
                                    [--launch-json '<kernel-like>' '<workload key>']

--warmup W --steps K: the traced command was `bench.py --warmup W --steps K` with the secondary legs switched off, so every
kernel ran (W + K) x its launches per step; the first W / (W + K) of every kernel's launches (in start order) are the
warm-up steps and are DROPPED, the table then covers the timed launches only (round 3's table included three warm-up
launches of the dominant kernel, and its average sat 8 % above the in-process HIP-event figure).  Kernels whose launch count
is not a multiple of W + K are kept whole and marked.
--launch-json: also write the dominant kernel's average over the timed launches into profiles/kernel_launch_ms.json
(bench.py reports it as roofline.launch_ms_rocprof next to its own HIP-event launch_ms)."""
import datetime
import json
import os
import sqlite3
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = 'name' if 'name' in cols else [c for c in cols if 'name' in c][0]
    per = {}
    for n, s, e in cur.execute("select %s, start, end from kernels order by start" % name_col):
        per.setdefault(n, []).append(e - s)
    return per


def main(argv):
    args = list(argv)
    warm = steps = 0
    like = wkey = None
    if '--warmup' in args:
        i = args.index('--warmup'); warm = int(args[i + 1]); del args[i:i + 2]
    if '--steps' in args:
        i = args.index('--steps'); steps = int(args[i + 1]); del args[i:i + 2]
    if '--launch-json' in args:
        i = args.index('--launch-json'); like, wkey = args[i + 1], args[i + 2]; del args[i:i + 3]
    path = args[0]
    out = args[1] if len(args) > 1 else None
    per = load(path)
    rows, notes = [], []
    for n, ds in per.items():
        kept, mark = ds, ''
        if warm > 0 and steps > 0:
            if len(ds) % (warm + steps) == 0:
                kept = ds[len(ds) * warm // (warm + steps):]
            else:
                mark = ' *'
        rows.append((n + mark, kept))
    tot = sum(sum(k) for _, k in rows) or 1
    head = '%-78s %8s %14s %12s %12s %12s %7s' % ('kernel', 'calls', 'total_ns', 'avg_ns', 'min_ns', 'max_ns', 'pct')
    lines = []
    if warm > 0 and steps > 0:
        lines.append('# timed launches only: the first %d of every %d launches of a kernel (the warm-up steps of bench.py --warmup %d '
                     '--steps %d) are dropped; "*" = launch count not a multiple of %d, kept whole' % (warm, warm + steps, warm, steps, warm + steps))
    lines.append(head)
    for n, k in sorted(rows, key=lambda r: -sum(r[1])):
        short = n if len(n) <= 78 else n[:75] + '...'
        lines.append('%-78s %8d %14d %12.0f %12d %12d %6.2f%%' % (short, len(k), sum(k), sum(k) / len(k), min(k), max(k), 100.0 * sum(k) / tot))
    text = '\n'.join(lines)
    print(text)
    if out:
        open(out, 'w').write(text + '\n')
    if like:
        sys.path.insert(0, REPO)
        from bench import kernel_source_hash
        hit = [(n, k) for n, k in rows if like in n]
        if not hit:
            raise SystemExit('no kernel like %r in %s' % (like, path))
        n, k = max(hit, key=lambda r: sum(r[1]))
        entry = {'kernel_like': like, 'workload': wkey, 'kernel': n[:160], 'launches': len(k),
                 'avg_ms_timed_launches': round(sum(k) / len(k) / 1e6, 4), 'min_ms': round(min(k) / 1e6, 4), 'max_ms': round(max(k) / 1e6, 4),
                 'source': 'rocprofv3 --kernel-trace --stats of bench.py --warmup %d --steps %d (secondary legs off), warm-up launches dropped' % (warm, steps),
                 'measured': datetime.date.today().isoformat(), 'kernel_source_sha256': kernel_source_hash()}
        jp = os.path.join(REPO, 'profiles', 'kernel_launch_ms.json')
        entries = json.load(open(jp)) if os.path.exists(jp) else []
        entries = [e for e in entries if (e.get('kernel_like'), e.get('workload')) != (like, wkey)] + [entry]
        json.dump(entries, open(jp, 'w'), indent=1)
        print(json.dumps(entry))


if __name__ == '__main__':
    main(sys.argv[1:])

"""one forward's kernel timeline from a rocprofv3 --kernel-trace database: python tools/diag/timeline.py <dir>"""
import glob, re, sqlite3, sys
db = sqlite3.connect(glob.glob(sys.argv[1] + '/**/*.db', recursive=True)[0])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = cur.execute(f'select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start').fetchall()
names = [re.sub(r'\(.*', '', r[0]).replace('gnnmp::', '').replace('void ', '') for r in rows]
idx = [i for i, x in enumerate(names) if x == names[-1]]
period = idx[-1] - idx[-2]
n = len(rows)
seg, sn = rows[n - 2 * period:n - period], names[n - 2 * period:n - period]
t0 = seg[0][1]
tot = 0
for (nm, s, e), x in zip(seg, sn):
    print('%-64s start %7.1f us  dur %6.1f us' % (x[:64], (s - t0) / 1e3, (e - s) / 1e3))
    tot += (e - s) / 1e3
print('kernels per forward %d, span %.1f us, kernel sum %.1f us' % (period, (seg[-1][2] - t0) / 1e3, tot))

#!/usr/bin/env python
"""profiles/kernel_mfma.json from a tools/pmc_passes.sh summary: the matrix-pipe FLOPs one step of a workload EXECUTES.

    python tools/mfma_json.py <summary.txt> --workload 'maze2 N=1000 k1=8 graphs=256 fp32' --steps-in-run 8

Per kernel: FLOPs per launch = 512 x (SQ_INSTS_VALU_MFMA_MOPS_F32 + _BF16 + _F16 + _F64 + _I8) when the MOPS counters are in the
summary (they count the multiplies and adds of every MFMA in units of 512, whatever the instruction's shape), else SQ_INSTS_MFMA x
--flops-per-mfma.  Per step = sum over kernels of FLOPs per launch x dispatches / --steps-in-run (bench.py --steps K --warmup W
runs W + 2K + 1 forwards: warm-up, timed, one profiled warm-up, profiled).  Entries are stamped like profiles/kernel_traffic.json
(workload, date, kernel source hash); bench.py turns them into whole_forward.executed_TFLOPs / frac_executed."""
import argparse, datetime, json, os, re, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bench import kernel_source_hash  # noqa: E402

MOPS = ('SQ_INSTS_VALU_MFMA_MOPS_F32', 'SQ_INSTS_VALU_MFMA_MOPS_BF16', 'SQ_INSTS_VALU_MFMA_MOPS_F16', 'SQ_INSTS_VALU_MFMA_MOPS_F64',
        'SQ_INSTS_VALU_MFMA_MOPS_I8')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('summary')
    ap.add_argument('--workload', required=True)
    ap.add_argument('--steps-in-run', type=int, required=True)
    ap.add_argument('--flops-per-mfma', type=float, default=0.0, help='fallback when the MOPS counters are absent')
    a = ap.parse_args()
    kern, cur = {}, None
    for ln in open(a.summary):
        if ln and not ln.startswith(' '):
            cur = ln.strip()
            continue
        m = re.match(r'\s+(\S+)\s+per-dispatch\s+([0-9.eE+-]+)\s+\(dispatches (\d+)', ln)
        if m and cur:
            kern.setdefault(cur, {})[m.group(1)] = (float(m.group(2)), int(m.group(3)))
    rows, total = [], 0.0
    for name, c in kern.items():
        have = [k for k in MOPS if k in c]
        if have:
            per_launch = 512.0 * sum(c[k][0] for k in have)
            n = c[have[0]][1]
            how = '512 x MOPS'
        elif 'SQ_INSTS_MFMA' in c and a.flops_per_mfma > 0:
            per_launch, n, how = c['SQ_INSTS_MFMA'][0] * a.flops_per_mfma, c['SQ_INSTS_MFMA'][1], 'SQ_INSTS_MFMA x %g' % a.flops_per_mfma
        else:
            continue
        if per_launch <= 0:
            continue
        per_step = per_launch * n / a.steps_in_run
        total += per_step
        rows.append({'kernel': name[:110], 'mfma_flops_per_launch': per_launch, 'launches_per_step': round(n / a.steps_in_run, 3),
                     'mfma_insts_per_launch': c.get('SQ_INSTS_MFMA', (None,))[0],
                     'mops': {k[len('SQ_INSTS_VALU_MFMA_MOPS_'):]: c[k][0] for k in have if c[k][0] > 0}, 'how': how})
    if not rows:
        raise SystemExit('no MFMA counters in %s' % a.summary)
    rows.sort(key=lambda r: -r['mfma_flops_per_launch'] * r['launches_per_step'])
    e = {'workload': a.workload, 'measured': datetime.date.today().isoformat(), 'kernel_source_sha256': kernel_source_hash(),
         'executed_mfma_flops_per_step': total, 'steps_in_run': a.steps_in_run,
         'how': 'rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_* passes (tools/pmc_passes.sh), x 512 FLOP, per-dispatch averages x '
                'launches per step, summed over the kernels of one forward', 'kernels': rows}
    path = os.path.join(REPO, 'profiles', 'kernel_mfma.json')
    old = json.load(open(path)) if os.path.exists(path) else []
    old = [o for o in old if o.get('workload') != a.workload]
    old.append(e)
    json.dump(old, open(path, 'w'), indent=1)
    print(json.dumps({k: v for k, v in e.items() if k != 'kernels'}))
    for r in rows:
        print('  %-110s %.4g FLOP/launch x %.2f' % (r['kernel'], r['mfma_flops_per_launch'], r['launches_per_step']))


if __name__ == '__main__':
    main()

#!/usr/bin/env python
"""Run ON THE GPU BOX: full-size parity census of the explorer forward (fp32 mode) against the fp64 oracle.

The golden fixtures pin 16 small graphs; the full-size tests used to compare 3 of 256 (cfg 2), 2 of 64 (cfg 3 shape) and 4 of
256 (cfg 4) graphs with the oracle.  This runs the oracle (oracle/ref_cpu.py, the reference's formulation of model.py:115-150,
in fp32 AND fp64) on EVERY graph of

    cfg 2        256 x maze2  1000-node k1 = 8      (BASELINE configs[1])
    cfg 3 shape   64 x kuka7  2000-node k1 = 10     (configs[2] shape, fp32 operands)
    cfg 5 shape    8 x kuka14 5000-node k1 = 16     (configs[4] shape, fp32 operands)
    cfg 4         16 per family of maze2 / snake7 / ur5 / kuka7, 1000-node k1 = 8   (configs[3])

and prints, per workload, the histogram over graphs of max|gpu - ref64| (the distance to the exact result), of the reference's
own fp32-vs-fp64 distance `own`, and of max|gpu - ref32|; the worst graphs; how many graphs exceed 1e-5 against fp64; and on
how many graphs the GPU is further from the exact result than the reference's own fp32 run.  `ref32` is the oracle's
MATERIALISING form (model.py:178-179 literally: the form that reproduces the reference's recorded fp32 scores bit for bit,
tests/test_oracle_golden.py), not the contracted one.  Both readings of north_star's "within 1e-5 fp32" are printed per
workload as ELEMENT counts: |gpu - ref32| > 1e-5, > 2e-5, and the failures of allclose(rtol = 1e-5, atol = 1e-5) against ref32
(the same three for the reference's own fp32 run against fp64 next to them, as the yardstick).
tests/test_parity_census_gpu.py asserts the per-workload bar through the same functions, incl. the ABSOLUTE ceilings below.

    python tools/parity_census.py [--quick] > profiles/r05_parity_census.txt
"""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))
import gnnmp  # noqa: E402
from gnnmp.synth import ENVS, synth_batch_gpu  # noqa: E402
from gnnmp.weights import load_weights  # noqa: E402

DEV = 'cuda:0'
WORKLOADS = [                                   # (name, env, nodes, k1, graphs, seed0)
    ('cfg2  maze2  N=1000 k1=8', 'maze2', 1000, 8, 256, 1234),
    ('cfg3  kuka7  N=2000 k1=10', 'kuka7', 2000, 10, 64, 1234),
    ('cfg5  kuka14 N=5000 k1=16', 'kuka14', 5000, 16, 8, 1234),
    ('cfg4  maze2  N=1000 k1=8', 'maze2', 1000, 8, 16, 5000),
    ('cfg4  snake7 N=1000 k1=8', 'snake7', 1000, 8, 16, 6000),
    ('cfg4  ur5    N=1000 k1=8', 'ur5', 1000, 8, 16, 7000),
    ('cfg4  kuka7  N=1000 k1=8', 'kuka7', 1000, 8, 16, 8000),
]
# Absolute ceilings per workload (round 5; VERDICT r4 weak #1: the quantile-matched bar alone let a 4x regression through at cfg 2):
# max over graphs of max|gpu - ref64| <= 1.25 x the maximum measured in round 4 (profiles/r04_parity_census.txt), the 7-DoF arm
# shapes at the bare north_star 1e-5; and the number of scores further than 1e-5 from fp64 <= 1.5 x the measured count + 5.
CEILINGS = {                                    # name: (max|gpu - ref64| ceiling, measured r4 max, elements > 1e-5 vs fp64 measured r4)
    'cfg2  maze2  N=1000 k1=8': (2.5e-5, 2.007e-5, 430),
    'cfg3  kuka7  N=2000 k1=10': (1.0e-5, 7.056e-6, 0),
    'cfg5  kuka14 N=5000 k1=16': (1.47e-5, 1.170e-5, 12),
    'cfg4  maze2  N=1000 k1=8': (1.85e-5, 1.477e-5, 29),
    'cfg4  snake7 N=1000 k1=8': (1.26e-5, 1.007e-5, 1),
    'cfg4  ur5    N=1000 k1=8': (2.28e-5, 1.823e-5, 136),
    'cfg4  kuka7  N=1000 k1=8': (1.0e-5, 5.322e-6, 0),
}
# workloads on which allclose(gpu, ref32, rtol = 1e-5, atol = 1e-5) holds on EVERY score today (profiles/r05_parity_census.txt);
# asserted to stay at zero failures
ALLCLOSE_HOLDS = {'cfg3  kuka7  N=2000 k1=10', 'cfg5  kuka14 N=5000 k1=16', 'cfg4  snake7 N=1000 k1=8', 'cfg4  ur5    N=1000 k1=8',
                  'cfg4  kuka7  N=1000 k1=8'}
# the 116-obstacle maze workloads: allclose failures against ref32 measured in round 5 (the reference's own fp32 run fails the same test
# against fp64 on 30 / 17 scores there); ceiling 1.5 x + 5 like the other counts
ALLCLOSE_MEASURED = {'cfg2  maze2  N=1000 k1=8': 37, 'cfg4  maze2  N=1000 k1=8': 13}
EDGES = [0.0, 1e-6, 2e-6, 4e-6, 6e-6, 8e-6, 1e-5, 1.5e-5, 2e-5, 3e-5, 1.0]


def census(env, nodes, k1, n_graphs, seed0, loop=5, mlp_dtype='fp32', device=DEV):
    """Per graph: (err64 = max|gpu - ref64|, err32 = max|gpu - ref32|, own = max|ref32 - ref64|, #elements > 1e-5 vs ref64, E).
    The GPU scores come from ONE batched forward over all graphs (the measured configuration)."""
    from oracle import ref_cpu
    torch.set_num_threads(min(16, torch.get_num_threads()))      # these graphs are small: a 128-thread pool only adds overhead
    e = ENVS[env]
    w = load_weights(e['ckpt'])
    w64 = {k: (t.double() if t.is_floating_point() else t) for k, t in w.items()}
    graphs = synth_batch_gpu(env, nodes, k1, n_graphs, device, seed0=seed0)
    m = gnnmp.EncoderProcessDecoder(e['workspace'], e['C'], e['d'], e['S']).eval()
    m.load_state_dict(w, strict=True)
    m.mlp_dtype = mlp_dtype
    b = gnnmp.GraphBatch.from_graphs(graphs, e['S'], device)
    parts = [p.cpu().double() for p in b.split_edges(m.forward_batch(b, loop))]
    rows = []
    for g, s in zip(graphs, parts):
        v, goal, obs, ei = (g[k].cpu() for k in ('v', 'goal', 'obstacles', 'edge_index'))
        r32 = ref_cpu.explorer_forward(w, v, goal, obs, ei, loop, materialize=True).double()      # the bit-exact pin of the reference
        r64 = ref_cpu.explorer_forward(w64, v.double(), goal.double(), obs.double(), ei, loop)
        d64, d32, o = (s - r64).abs(), (s - r32).abs(), (r32 - r64).abs()
        rows.append(Row((d64.max().item(), d32.max().item(), o.max().item(), int((d64 > 1e-5).sum()), int(ei.shape[1])),
                        n32_1e5=int((d32 > 1e-5).sum()), n32_2e5=int((d32 > 2e-5).sum()),
                        n32_allclose=int((d32 > 1e-5 + 1e-5 * r32.abs()).sum()),
                        own_1e5=int((o > 1e-5).sum()), own_2e5=int((o > 2e-5).sum()), own_allclose=int((o > 1e-5 + 1e-5 * r64.abs()).sum())))
    return rows


class Row(tuple):
    """(err64, err32, own, #elements > 1e-5 vs ref64, E) of one graph + the element counts of both north_star readings."""
    def __new__(cls, t, **counts):
        r = super().__new__(cls, t)
        r.counts = counts
        return r


def hist(vals):
    out = []
    for lo, hi in zip(EDGES[:-1], EDGES[1:]):
        n = sum(1 for x in vals if lo <= x < hi)
        out.append(n)
    return out


def report(name, rows):
    n = len(rows)
    print('%s: %d graphs, %d edge scores' % (name, n, sum(r[4] for r in rows)))
    print('    %-28s' % 'bin (upper edge)' + ''.join('%9s' % ('<%.1e' % hi if hi < 1 else '>=3e-5') for hi in EDGES[1:]))
    for label, col in (('max|gpu - ref64| per graph', 0), ('own = max|ref32 - ref64|', 2), ('max|gpu - ref32| per graph', 1)):
        print('    %-28s' % label + ''.join('%9d' % c for c in hist([r[col] for r in rows])))
    worst = sorted(range(n), key=lambda i: -rows[i][0])[:3]
    print('    worst graphs vs fp64: ' + '; '.join('#%d %.2e (own %.2e, vs ref32 %.2e)' % (i, rows[i][0], rows[i][2], rows[i][1]) for i in worst))
    over = [i for i in range(n) if rows[i][0] > 1e-5]
    print('    max over graphs: |gpu - ref64| %.3e   own %.3e   |gpu - ref32| %.3e;  graphs over 1e-5 vs fp64: %d of %d (%d of %d elements)' % (
        max(r[0] for r in rows), max(r[2] for r in rows), max(r[1] for r in rows), len(over), n, sum(r[3] for r in rows), sum(r[4] for r in rows)))
    st = stats(rows)
    print('    medians: |gpu - ref64| %.3e   own %.3e;  graphs on which the GPU is FURTHER from fp64 than the reference\'s own fp32 run: %d of %d' % (
        st['med_err64'], st['med_own'], st['n_worse_than_ref'], n))
    print('    quantile-matched bar max(1e-5, 1.25 x own): ' + '; '.join('%s %.3e <= %.3e %s' % (k, st['errs'][k], st['bars'][k], 'ok' if st['errs'][k] <= st['bars'][k] else 'EXCEEDED')
                                                                         for k in ('median', 'p90', 'max')))
    tot = {k: sum(r.counts[k] for r in rows) for k in rows[0].counts}
    ne = sum(r[4] for r in rows)
    print('    elementwise, of %d scores:  |gpu - ref32| > 1e-5: %d   > 2e-5: %d   allclose(rtol 1e-5, atol 1e-5) failures vs ref32: %d' % (
        ne, tot['n32_1e5'], tot['n32_2e5'], tot['n32_allclose']))
    print('    yardstick (the reference\'s own fp32 run against fp64):  |ref32 - ref64| > 1e-5: %d   > 2e-5: %d   allclose failures: %d' % (
        tot['own_1e5'], tot['own_2e5'], tot['own_allclose']))
    if name in CEILINGS:
        c = CEILINGS[name]
        print('    absolute ceiling: max|gpu - ref64| %.3e <= %.3e %s;  elements over 1e-5 vs fp64 %d <= %d %s' % (
            st['max_err64'], c[0], 'ok' if st['max_err64'] <= c[0] else 'EXCEEDED', sum(r[3] for r in rows), int(1.5 * c[2]) + 5,
            'ok' if sum(r[3] for r in rows) <= int(1.5 * c[2]) + 5 else 'EXCEEDED'))
    return len(over)


def totals(rows):
    return {k: sum(r.counts[k] for r in rows) for k in rows[0].counts}


def stats(rows):
    """Population-level figures of one workload.  The reference's own fp32-vs-fp64 distance `own` is a property of the WORKLOAD
    (graph size, in-degree, obstacle count) whose per-graph maximum over ~10^4..10^5 scores fluctuates several-fold between graphs
    of one workload, so the bar of tests/parity_bar.py, max(1e-5, 1.25 own), is applied QUANTILE BY QUANTILE: the median, the
    90th percentile and the maximum over graphs of max|gpu - ref64| must each stay below max(1e-5, 1.25 x the same quantile of
    own) -- the GPU's error distribution is dominated by the reference's own rounding-noise distribution."""
    e = sorted(r[0] for r in rows)
    o = sorted(r[2] for r in rows)
    n = len(rows)
    at = lambda x, q: x[min(n - 1, int(q * n))]  # noqa: E731
    qs = {'median': 0.5, 'p90': 0.9, 'max': 1.0}
    bars = {k: max(1e-5, 1.25 * at(o, q)) for k, q in qs.items()}
    errs = {k: at(e, q) for k, q in qs.items()}
    return {'med_err64': at(e, 0.5), 'med_own': at(o, 0.5), 'max_err64': e[-1], 'max_own': o[-1], 'bars': bars, 'errs': errs,
            'bar': bars['max'], 'n_worse_than_ref': sum(1 for r in rows if r[0] > r[2]),
            'n_over_bar': sum(1 for k in qs if errs[k] > bars[k]), 'ok': all(errs[k] <= bars[k] for k in qs)}


def main():
    quick = '--quick' in sys.argv
    print('explorer forward, fp32 mode, every graph of each workload against the fp64 oracle (oracle/ref_cpu.py); north_star bar 1e-5')
    bad = 0
    t0 = time.time()
    for name, env, nodes, k1, ng, seed0 in WORKLOADS:
        rows = census(env, nodes, k1, max(2, ng // 8) if quick else ng, seed0)
        bad += report(name, rows)
    print('census done in %.0f s; graphs over 1e-5 against fp64: %d' % (time.time() - t0, bad))
    return bad


if __name__ == '__main__':
    sys.exit(1 if main() else 0)

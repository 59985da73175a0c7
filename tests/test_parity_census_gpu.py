"""Full-size parity census (tools/parity_census.py): the fp64 oracle on EVERY graph of the BASELINE workloads -- all 256 cfg-2
graphs, all 64 graphs of the cfg-3 shape (fp32 operands), 8 graphs of the cfg-5 shape, 16 per family of cfg 4.

What is asserted, per workload (profiles/r05_parity_census.txt has the histograms and the elementwise counts of both readings of
"within 1e-5 fp32"; ref32 is the oracle's materialising form, the bit-exact pin of the reference):
  * ABSOLUTE ceilings per workload (tools/parity_census.py CEILINGS): max|gpu - ref64| <= 1.25 x round 4's measured maximum
    (2.5e-5 at cfg 2; the 7-DoF arm shapes at the bare 1e-5), the count of scores beyond 1e-5 <= 1.5 x measured + 5, and zero
    allclose(1e-5, 1e-5) failures against ref32 wherever that holds today;
  * the quantile-matched form of tests/parity_bar.py's bar: the median, the 90th percentile and the maximum over graphs of
    max|gpu - ref64| each stay below max(1e-5, 1.25 x the same quantile of own), own = max|ref32 - ref64| of a graph = how far the
    reference's own fp32 run is from the exact result (its per-graph maximum over 10^4 .. 10^5 scores fluctuates several-fold
    between graphs of one workload, so one graph's own is a noisy yardstick for that graph);
  * the bare north_star figure 1e-5 against fp64 wherever the reference's own fp32 run holds it on every graph of the workload
    (max own <= 1e-5: the 7-DoF arm workloads);
  * where the reference's noise is above the 1e-5 floor (median own > 1e-5: the 116-obstacle mazes, ur5), the GPU's worst graph is
    better than the reference's worst graph.
On the 116-obstacle maze workloads the bare 1e-5 against fp64 does NOT hold on every graph (round 4 census: 96 of 256 cfg-2 graphs
between 1.0e-5 and 2.0e-5 -- 430 of 2.9 M scores -- where the reference's own fp32 run is 1.5e-5 ... 6.2e-5 away and further from
fp64 than the GPU on 255 of the 256 graphs): a CPU experiment (tools/diag/parity_upgrade.py) shows that even the whole node side in
double precision leaves 1.1e-5 -- the remaining distance is fp32 rounding spread over every stage, not one amplified stretch.
Slow: about two minutes of CPU oracle time."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
import parity_census  # noqa: E402

pytestmark = [pytest.mark.gpu, pytest.mark.slow]


@pytest.mark.parametrize('name,env,nodes,k1,n_graphs,seed0', parity_census.WORKLOADS, ids=[w[0].replace(' ', '_') for w in parity_census.WORKLOADS])
def test_every_graph_within_the_workload_bar(name, env, nodes, k1, n_graphs, seed0):
    rows = parity_census.census(env, nodes, k1, n_graphs, seed0)
    assert len(rows) == n_graphs
    parity_census.report(name, rows)
    st = parity_census.stats(rows)
    assert st['ok'], '%s: max|gpu - ref64| over the quantile-matched bar: %s vs %s' % (name, st['errs'], st['bars'])
    if st['max_own'] <= 1e-5:
        assert st['max_err64'] <= 1e-5, '%s: the reference holds 1e-5 against fp64 on every graph, the GPU does not (%.3e)' % (name, st['max_err64'])
    # absolute ceilings (round 5): the worst score of the workload against fp64 stays within 1.25 x what round 4 measured (the arm
    # shapes at the bare 1e-5) and the count of scores beyond 1e-5 does not grow past 1.5 x + 5; a rounding regression that the
    # quantile-matched bar would let through (it is 7.8e-5 at cfg 2 against 2.0e-5 measured) fails here
    ceil, _, n_over = parity_census.CEILINGS[name]
    scale = n_graphs / {w[0]: w[4] for w in parity_census.WORKLOADS}[name]
    assert st['max_err64'] <= ceil, '%s: max|gpu - ref64| %.3e over the absolute ceiling %.3e' % (name, st['max_err64'], ceil)
    assert sum(r[3] for r in rows) <= int(1.5 * n_over * scale) + 5, (name, sum(r[3] for r in rows), n_over)
    if name in parity_census.ALLCLOSE_HOLDS:
        assert parity_census.totals(rows)['n32_allclose'] == 0, (name, parity_census.totals(rows))
    else:
        assert parity_census.totals(rows)['n32_allclose'] <= int(1.5 * parity_census.ALLCLOSE_MEASURED[name] * scale) + 5, (name, parity_census.totals(rows))
    if st['med_own'] > 1e-5:        # (below the 1e-5 floor both are rounding noise of the same size; nothing to rank)
        assert st['max_err64'] <= st['max_own'], (name, st)

"""The call the reference itself makes (eval_gnn.py:193-196: obs_data + H2D + model(**kw) + .cpu()) at the published planner
setting -- N = 1002 nodes, k = 30 (k1 = 41) -- stays a one-millisecond call, also right after the caching allocator was emptied
and with tens of milliseconds of host work between calls (cold host caches, the planner's rhythm).  BENCH_r05 printed 5.0 ms per
problem for this span while its kernels take 0.24 ms: a 128-thread torch pool spinning inside a 16-CPU cgroup quota got the
process throttled (gnnmp/hostenv.py); the host loop now runs with the pool inside the quota and the span is bounded here."""
import os
import time

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _setup():
    import gnnmp
    from gnnmp import planner
    from gnnmp.graph_build import create_data
    from gnnmp.maze2d import Maze2D
    from gnnmp.weights import load_weights
    with np.load(os.path.join(GOLDEN, 'evalset_mazehard_first1000.npz')) as f:
        env = Maze2D(f['maps'], f['init_states'], f['goal_states'])
    m = gnnmp.EncoderProcessDecoder(2, 2, 32, 2).eval()
    m.load_state_dict(load_weights('weights_maze'))
    np.random.seed(4321)
    env.init_new_problem(0)
    free, collided = env.sample_n_points(500, need_negative=True)
    collided = collided[:len(free)]
    free = [env.init_state] + [env.goal_state] + list(free)
    data = create_data(free, collided, env.goal_state, 30)
    return env, m, planner, free, collided, data


def _span(env, m, planner, free, collided, data, dev, reference_kwargs):
    t = time.perf_counter()
    od = planner.obs_data(env, free, collided, dev, obstacles_only=not reference_kwargs)
    kw = dict(goal=data['goal'].to(dev), v=data['v'].to(dev), edge_index=data['edge_index'].to(dev), loop=5, **od)
    if reference_kwargs:
        kw['labels'] = data['labels'].to(dev)
    P = m(**kw).detach().cpu().numpy()
    return time.perf_counter() - t, P


@pytest.mark.parametrize('reference_kwargs', [False, True])
def test_dropin_forward_span_is_one_millisecond(reference_kwargs):
    from gnnmp.hostenv import limit_host_threads
    dev = torch.device('cuda:0')
    env, m, planner, free, collided, data = _setup()
    assert data['v'].shape[0] == 1002
    threads0 = torch.get_num_threads()
    limit_host_threads(1)                                     # the host loop is a one-core loop (bench.py planner_leg does the same)
    try:
        _, P0 = _span(env, m, planner, free, collided, data, dev, reference_kwargs)           # first use: handle, workspace, code objects
        medians = []
        for _ in range(3):
            torch.cuda.empty_cache()                          # workspace of the module survives (it is referenced); the 4 MB blocks do not
            ts = []
            for i in range(15):
                t_busy = time.perf_counter()                  # ~10 ms of one-core host work between calls (the planner's rhythm; no sleep:
                while time.perf_counter() - t_busy < 0.01:    # an idle core and an idle GPU clock down, which is not what is bounded here)
                    junk = [j * j for j in range(2000)]
                dt, P = _span(env, m, planner, free, collided, data, dev, reference_kwargs)
                ts.append(dt)
                assert np.array_equal(P, P0)
            medians.append(sorted(ts)[len(ts) // 2])
    finally:
        torch.set_num_threads(threads0)
    print('drop-in forward span (N = 1002, k1 = 41), reference_kwargs=%s: medians of 3 x 15 calls %s ms' % (reference_kwargs, [round(x * 1e3, 3) for x in medians]))
    # measured on MI355X: 0.55-0.8 ms (obstacles only), 0.7-0.95 ms (every reference keyword); the kernels are 0.24 ms of it
    assert min(medians) <= 1.0e-3, 'drop-in forward span: medians %s ms' % [round(x * 1e3, 3) for x in medians]

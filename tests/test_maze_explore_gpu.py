"""Explore stage fully on the device for 2-D mazes (maze_kernels.hip: greedy best-edge expansion + grid
collision checker, one wavefront per problem; graphs from graph_kernels.hip; batched explorer forward) against
the outcomes of the reference planner at its default configuration (tests/golden/evalset_*.npz) and against the
host counterpart (planner.explore) decision by decision."""
import numpy as np
import pytest
import torch

from conftest import golden_files, load_weights
import gnnmp
from gnnmp import planner
from gnnmp.maze2d import Maze2D

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _models():
    m = gnnmp.EncoderProcessDecoder(2, 2, 32, 2).eval()
    m.load_state_dict(load_weights('weights_maze'))
    return m


def test_device_explore_matches_reference_outcomes():
    with np.load(golden_files('evalset_mazehard_first12')[0]) as f:
        r = {k: f[k] for k in f.files}
    n = r['rows'].shape[0]
    problems = [dict(map=r['maps'][i], init_state=r['init_states'][i], goal_state=r['goal_states'][i]) for i in range(n)]
    np.random.seed(int(r['seed']))
    torch.manual_seed(int(r['seed']))
    res = planner.explore_maze_batch(problems, _models(), DEV, batch=int(r['batch']), k=int(r['k']))
    ref = r['rows']
    got_c = [x['c_explore'] for x in res]
    print('\nc_explore device %s\n          ref    %s' % (got_c, ref[:, 3].astype(int).tolist()))
    assert [int(x['success']) for x in res] == ref[:, 0].astype(int).tolist()
    assert got_c == ref[:, 3].astype(int).tolist()                        # collision checks of the explore stage
    assert [len(x['explored']) for x in res] == ref[:, 6].astype(int).tolist()
    assert [len(x['path']) for x in res] == ref[:, 5].astype(int).tolist()
    assert np.allclose([planner.path_cost(x['path']) for x in res], ref[:, 1], rtol=0, atol=1e-9)


def test_device_explore_equals_host_counterpart_step_by_step():
    """Same problems through planner.explore (host frontier + host collision checker): identical explored
    order, identical explored_edges list (incl. the initial [0, 0]), identical path."""
    with np.load(golden_files('evalset_mazehard_first12')[0]) as f:
        r = {k: f[k] for k in f.files}
    idx = [0, 3, 5]
    m = _models()
    ms = gnnmp.ModelSmoother(workspace_size=2, config_size=2, embed_size=128, obs_size=6).eval()
    ms.load_state_dict(load_weights('smooth_2d_attv3'))
    problems = [dict(map=r['maps'][i], init_state=r['init_states'][i], goal_state=r['goal_states'][i]) for i in idx]
    np.random.seed(7)
    dev_res = planner.explore_maze_batch(problems, m, DEV, batch=300, k=20)
    np.random.seed(7)
    # the host path samples problem by problem in the same order -> same numpy stream
    host = []
    envs = []
    for i in idx:
        env = Maze2D(r['maps'][i][None], r['init_states'][i][None], r['goal_states'][i][None])
        env.init_new_problem(0)
        envs.append(env)
    # sample first for all (like the batched path), then run the host greedy loops on the same samples
    samples = []
    for env in envs:
        free, coll = env.sample_n_points(300, need_negative=True)
        samples.append((free, coll[:len(free)]))
    for env, (free, coll), d in zip(envs, samples, dev_res):
        free = [env.init_state] + [env.goal_state] + list(free)
        data = planner.create_data(free, coll, env.goal_state, 20)
        od = planner.obs_data(env, free, coll, DEV)
        sc = m.edge_scores(goal=data['goal'].to(DEV), v=data['v'].to(DEV), edge_index=data['edge_index'].to(DEV), loop=5,
                           obstacles=od['obstacles']).cpu().numpy()
        state = {'explored': [0], 'explored_edges': [[0, 0]], 'costs': {0: 0.}, 'prev': {0: 0}}
        path = planner.greedy_expand_sparse(sc, data['edge_index'].numpy(), data['labels'].numpy(), data['v'].numpy(), env, state)
        assert d['explored'].tolist() == state['explored']
        assert d['explored_edges'].tolist() == state['explored_edges']
        assert d['success'] == (path is not None)
        if path is not None:
            assert [tuple(x) for x in d['path']] == [tuple(data['v'][j].numpy()) for j in path]
        assert d['c_explore'] == env.collision_check_count


def _smoother():
    ms = gnnmp.ModelSmoother(workspace_size=2, config_size=2, embed_size=128, obs_size=6).eval()
    ms.load_state_dict(load_weights('smooth_2d_attv3'))
    return ms


def test_device_smoothing_matches_reference_outcomes():
    """Whole planner on the device (explore + 5 x (smoother forward, collision-checked steering)): per-problem
    collision-check counts of BOTH stages and the smoothed path cost against the reference planner's own run."""
    with np.load(golden_files('evalset_mazehard_first12')[0]) as f:
        r = {k: f[k] for k in f.files}
    n = r['rows'].shape[0]
    problems = [dict(map=r['maps'][i], init_state=r['init_states'][i], goal_state=r['goal_states'][i]) for i in range(n)]
    np.random.seed(int(r['seed']))
    torch.manual_seed(int(r['seed']))
    res = planner.explore_maze_batch(problems, _models(), DEV, batch=int(r['batch']), k=int(r['k']), model_s=_smoother())
    ref = r['rows']
    c_s = [x['c_smooth'] for x in res]
    cost = [planner.path_cost(x['smooth_path']) for x in res]
    print('\nc_smooth device %s\n         ref    %s' % (c_s, ref[:, 4].astype(int).tolist()))
    print('smooth cost device %s\n            ref    %s' % (np.round(cost, 5).tolist(), np.round(ref[:, 2], 5).tolist()))
    assert [x['c_explore'] for x in res] == ref[:, 3].astype(int).tolist()
    # same bar as the host-steered planner (tests/test_planner_evalset_gpu.py): identical steering decisions unless a
    # proposal sits within fp32 noise of the RRT_EPS threshold
    assert np.allclose(cost, ref[:, 2], rtol=1e-3, atol=1e-3)
    assert np.abs(np.array(c_s) - ref[:, 4]).max() <= 0.03 * ref[:, 4].max()


def test_device_smoothing_equals_host_steering_bitwise():
    """Device steering against planner.model_smooth (host steering with the Maze2D checker, same GPU smoother
    network): identical waypoints bit for bit and identical collision-check counts, on real solved problems and
    on a batch with a two-waypoint path (nothing to steer)."""
    with np.load(golden_files('evalset_mazehard_first12')[0]) as f:
        r = {k: f[k] for k in f.files}
    idx = [0, 2, 4, 8, 9]
    m, ms = _models(), _smoother()
    problems = [dict(map=r['maps'][i], init_state=r['init_states'][i], goal_state=r['goal_states'][i]) for i in idx]
    np.random.seed(11)
    res = planner.explore_maze_batch(problems, m, DEV, batch=400, k=25, model_s=ms)
    for d in res:
        if not d['success']:
            continue
        env = d['env']
        v = d['v'].numpy()
        # the samples the batched path handed to the smoother: free rows first (init, goal, samples), then collided
        n_free = d['n_free']
        free = [x for x in v[:n_free]]
        coll = [x for x in v[n_free:]]
        c0 = env.collision_check_count
        host = planner.model_smooth(ms, free, coll, [p.copy() for p in d['path']], env, DEV)      # list of float32 rows
        c_host = env.collision_check_count - c0
        assert len(host) == len(d['smooth_path'])
        assert all(np.array_equal(a, b) for a, b in zip(host, d['smooth_path'])), \
            np.abs(np.array(host) - np.array(d['smooth_path'])).max()
        assert c_host == d['c_smooth']


@pytest.mark.parametrize('seed', range(6))
def test_device_frontier_tie_breaking_on_quantised_scores(seed):
    """gnnmp_maze_explore fed with synthetic scores that take only a few distinct values (ties in every row, exact
    zeros, a few collided nodes): the cached-row-maximum frontier must pick cells in exactly the order of the host
    frontier (row-major first maximum over explored rows) -- same explored order, pair list, path and check counts."""
    from gnnmp import graph_build
    rng = np.random.RandomState(100 + seed)
    B = 5
    maps = (rng.rand(B, 15, 15) < 0.25).astype(np.float64)
    envs, vs, n_free, eis, scs = [], [], [], [], []
    np.random.seed(seed)
    for b in range(B):
        free_cells = np.argwhere(maps[b] == 0)
        cell2state = lambda c: (c + 0.5) / 15 * 2 - 1          # noqa: E731
        init, goal = cell2state(free_cells[rng.randint(len(free_cells))]), cell2state(free_cells[rng.randint(len(free_cells))])
        env = Maze2D(maps[b][None], init[None], goal[None])
        env.init_new_problem(0)
        free, coll = env.sample_n_points(int(rng.randint(30, 90)), need_negative=True)
        coll = coll[:len(free)]
        free = [env.init_state, env.goal_state] + list(free)
        data = planner.create_data(free, coll, env.goal_state, int(rng.randint(4, 12)))
        ei = data['edge_index'].numpy()
        sc = rng.choice(np.array([-2.1, -1.3, 0.0, 0.5, 0.5, 1.7], dtype=np.float32), size=ei.shape[1])
        envs.append(env); vs.append(data['v']); n_free.append(len(free)); eis.append(data['edge_index']); scs.append(sc)
    ptr = torch.zeros(B + 1, dtype=torch.int64)
    ptr[1:] = torch.tensor([x.shape[0] for x in vs]).cumsum(0)
    eptr = torch.zeros(B + 1, dtype=torch.int64)
    eptr[1:] = torch.tensor([x.shape[1] for x in eis]).cumsum(0)
    goal64 = torch.tensor(np.asarray([e.goal_state for e in envs], dtype=np.float64)).to(DEV)
    out = planner.maze_explore_device(torch.cat(vs).to(DEV), ptr.to(torch.int32).to(DEV), eptr.to(torch.int32).to(DEV), n_free,
                                      torch.cat(eis, dim=1).to(DEV), torch.from_numpy(np.concatenate(scs)).to(DEV),
                                      torch.tensor(maps).to(DEV), goal64)
    success, n_expl, n_pairs, plen, checks, expl, ee, ee_off, path = out
    nptr = ptr.tolist()
    for b in range(B):
        env = envs[b]
        c0 = env.collision_check_count
        labels = np.zeros((vs[b].shape[0], 2), dtype=np.int64)
        labels[n_free[b]:, 1] = 1
        state = {'explored': [0], 'explored_edges': [[0, 0]], 'costs': {0: 0.}, 'prev': {0: 0}}
        ref = planner.greedy_expand_sparse(scs[b], eis[b].numpy(), labels, vs[b].numpy(), env, state)
        assert expl[nptr[b]:nptr[b] + n_expl[b]].tolist() == state['explored'], b
        assert ee[ee_off[b]:ee_off[b + 1]].reshape(-1, 2).tolist() == state['explored_edges'], b
        assert bool(success[b]) == (ref is not None)
        if ref is not None:
            assert path[nptr[b]:nptr[b] + plen[b]].tolist() == ref
        assert checks[b] == env.collision_check_count - c0


def test_unsolvable_problem_matches_host_planner():
    """Goal cell walled in: the device planner must exhaust the frontier exactly like the host loop (same explored
    order, pair list and check count), report failure, and skip smoothing for that problem while its solvable batch
    neighbour is smoothed."""
    with np.load(golden_files('evalset_mazehard_first12')[0]) as f:
        r = {k: f[k] for k in f.files}
    m, ms = _models(), _smoother()
    walled = np.zeros((15, 15))
    walled[9:12, 9:12] = 1
    walled[10, 10] = 0                                                   # free cell enclosed by obstacles
    cell = lambda i, j: (np.array([i, j]) + 0.5) / 15 * 2 - 1            # noqa: E731
    problems = [dict(map=walled, init_state=cell(2, 2), goal_state=cell(10, 10)),
                dict(map=r['maps'][1], init_state=r['init_states'][1], goal_state=r['goal_states'][1])]
    np.random.seed(3)
    res = planner.explore_maze_batch(problems, m, DEV, batch=200, k=15, model_s=ms)
    assert not res[0]['success'] and len(res[0]['path']) == 0 and len(res[0]['smooth_path']) == 0 and res[0]['c_smooth'] == 0
    assert res[1]['success'] and len(res[1]['smooth_path']) == len(res[1]['path'])
    # host frontier on the same samples and the same scores
    d = res[0]
    env = Maze2D(walled[None], cell(2, 2)[None], cell(10, 10)[None])
    env.init_new_problem(0)
    v, nf = d['v'].numpy(), d['n_free']
    data = planner.create_data([x for x in v[:nf]], [x for x in v[nf:]], env.goal_state, 15)
    od = planner.obs_data(env, [x for x in v[:nf]], [x for x in v[nf:]], DEV)
    sc = m.edge_scores(goal=data['goal'].to(DEV), v=data['v'].to(DEV), edge_index=data['edge_index'].to(DEV), loop=5,
                       obstacles=od['obstacles']).cpu().numpy()
    state = {'explored': [0], 'explored_edges': [[0, 0]], 'costs': {0: 0.}, 'prev': {0: 0}}
    assert planner.greedy_expand_sparse(sc, data['edge_index'].numpy(), data['labels'].numpy(), data['v'].numpy(), env, state) is None
    assert d['explored'].tolist() == state['explored']
    assert d['explored_edges'].tolist() == state['explored_edges']
    assert d['c_explore'] - d['env'].collision_check_count == env.collision_check_count


@pytest.mark.parametrize('batch,k,density,seed', [(150, 12, 0.2, 0), (260, 18, 0.4, 1), (620, 10, 0.3, 2)])
def test_device_planner_equals_host_planner_on_random_maps(batch, k, density, seed):
    """Random occupancy maps (not from the reference's data set; the dense ones contain unsolvable problems), other
    batch / k settings: every problem through the device planner and through planner.explore (host frontier, host
    collision checks, host steering) -- same success flag, collision-check counts of both stages, explored order
    and bit-identical smoothed path.  (A 200-problem version of this loop was run once per shape with 0 mismatches.)
    The third shape has ~1240 nodes per problem: beyond the 1024 the explore kernel keeps in LDS, i.e. its instantiation
    with the per-node state in the workspace."""
    rng = np.random.RandomState(2024 + seed)
    B = 14
    maps = (rng.rand(B, 15, 15) < density).astype(np.float64)
    cell = lambda x: (x + 0.5) / 15 * 2 - 1                               # noqa: E731
    inits, goals = [], []
    for b in range(B):
        free_cells = np.argwhere(maps[b] == 0)
        inits.append(cell(free_cells[rng.randint(len(free_cells))]))
        goals.append(cell(free_cells[rng.randint(len(free_cells))]))
    inits, goals = np.array(inits), np.array(goals)
    m, ms = _models(), _smoother()
    problems = [dict(map=maps[b], init_state=inits[b], goal_state=goals[b]) for b in range(B)]
    np.random.seed(100 + seed)
    res = planner.explore_maze_batch(problems, m, DEV, batch=batch, k=k, model_s=ms)
    np.random.seed(100 + seed)
    env = Maze2D(maps, inits, goals)
    for b in range(B):
        env.init_new_problem(b)
        r = planner.explore(env, m, ms, True, batch=batch, t_max=batch, k=k, device=DEV, sparse=True)
        d = res[b]
        assert bool(r['success']) == d['success'], b
        assert r['c_explore'] == d['c_explore'] and r['explored'] == d['explored'].tolist(), b
        if r['success']:
            assert r['c_smooth'] == d['c_smooth'], b
            assert np.array_equal(np.array(r['smooth_path']), np.array(d['smooth_path'])), b

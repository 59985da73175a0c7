"""Host-side planner counterparts against traces recorded from the
reference planner on real MazeEnv problems (tools/gen_golden.py planner_cases).  CPU only: the GNNs
are replaced by replay objects that (a) assert they receive exactly the tensors the reference's
models received and (b) return the reference's recorded outputs, so the test pins
  H1 create_data (identical v / coalesced edge_index),
  H3 obs_data (identical free / collided / obstacles tensors),
  H2 masking + greedy expansion incl. the legacy-index quirk (identical explored, explored_edges, path,
     collision-check count),
  H4 model_smooth + steering (identical smoothed path and collision-check count),
and the Maze2D environment (identical sample stream from the seeded numpy RNG, identical check counts).
"""
import os

import numpy as np
import pytest
import torch

from conftest import golden_files
import gnnmp  # noqa: F401
from gnnmp import planner
from gnnmp.maze2d import Maze2D


def _load(path):
    with np.load(path) as f:
        return {k: f[k] for k in f.files}


class ReplayExplorer:
    def __init__(self, rec):
        self.rec, self.i = rec, 0

    def __call__(self, goal, v, labels, edge_index, loop, free, collided, obstacles):
        r, i = self.rec, self.i
        assert loop == 5
        assert np.array_equal(v.numpy(), r['e%d_v' % i])
        assert np.array_equal(edge_index.numpy(), r['e%d_edge_index' % i])
        assert np.array_equal(free.numpy(), r['e%d_free' % i])
        assert np.array_equal(collided.numpy(), r['e%d_collided' % i])
        assert np.array_equal(obstacles.numpy(), r['e%d_obstacles' % i])
        assert np.array_equal(goal.numpy(), r['e%d_goal' % i])
        n = v.shape[0]
        assert labels.shape == (n, 3) and float(labels[1, 2]) == 1.0
        P = torch.zeros(n, n)
        P[edge_index[1], edge_index[0]] = torch.from_numpy(r['e%d_scores' % i])
        self.i += 1
        return P


    def edge_scores(self, goal, v, labels, edge_index, loop, free, collided, obstacles):
        P = self(goal, v, labels, edge_index, loop, free, collided, obstacles)
        return P[edge_index[1], edge_index[0]]


class ReplaySmoother:
    def __init__(self, rec):
        self.rec, self.i = rec, 0

    def __call__(self, path, free, collided, obstacles, edge_index, loop):
        r, i = self.rec, self.i
        assert loop == 1
        assert np.array_equal(path.numpy(), r['s%d_path' % i])
        assert np.array_equal(free.numpy(), r['s%d_free' % i])
        assert np.array_equal(collided.numpy(), r['s%d_collided' % i])
        P = path.shape[0]
        assert edge_index.shape == (2, 3 * P - 2)
        self.i += 1
        return torch.from_numpy(r['s%d_out' % i])


def _env(r):
    env = Maze2D(r['map'][None], r['init_state'][None], r['goal_state'][None])
    env.init_new_problem(0)
    return env


@pytest.mark.parametrize('sparse', [False, True], ids=['dense', 'sparse_frontier'])
@pytest.mark.parametrize('path', golden_files('planner_'), ids=os.path.basename)
def test_planner_replay_matches_reference_trace(path, sparse):
    r = _load(path)
    env = _env(r)
    np.random.seed(int(r['seed']))
    torch.manual_seed(int(r['seed']))
    ex, sm = ReplayExplorer(r), ReplaySmoother(r)
    res = planner.explore(env, ex, sm, True, batch=int(r['batch']), t_max=int(r['t_max']), k=int(r['k']), device='cpu',
                          sparse=sparse)
    assert ex.i == int(r['n_forward']) and sm.i == int(r['n_smooth'])
    assert res['success'] == bool(r['success'])
    assert res['explored'] == r['explored'].tolist()
    assert res['explored_edges'] == r['explored_edges'].tolist()
    assert res['c_explore'] == int(r['c_explore'])
    assert res['c_smooth'] == int(r['c_smooth'])
    assert np.array_equal(np.array(res['path'], dtype=np.float32), r['path'])
    assert np.array_equal(np.array(res['smooth_path'], dtype=np.float64), r['smooth_path'])


def test_maze_env_counts_and_obstacles():
    r = _load(golden_files('planner_mazehard_0')[0])
    env = _env(r)
    assert env.obstacles.shape == (int(r['map'].sum()), 2)
    assert np.array_equal(env.obstacles.astype(np.float32), r['e0_obstacles'])
    # out-of-bounds queries are refused without being counted (maze_env.py:281-288)
    c = env.collision_check_count
    assert env._state_fp(np.array([1.5, 0.0])) is False and env.collision_check_count == c
    assert env._edge_fp(np.array([0.0, 0.0]), np.array([0.0, 2.0])) is False and env.collision_check_count == c
    # a zero-length edge costs exactly two point queries when free
    free_pt = env.init_state
    env._edge_fp(free_pt, free_pt)
    assert env.collision_check_count == c + 2


def test_legacy_index_quirk_is_reproduced():
    """explored_edges pairs are reshaped (2, -1), not transposed, and used as a (rows, cols) tuple."""
    P = np.ones((4, 4), dtype=np.float32)
    labels = np.zeros((4, 3), dtype=np.float32)
    planner._mask_policy(P, labels, [0], [[0, 0], [0, 2], [2, 0], [0, 3], [3, 0]])
    # flat list 0,0,0,2,2,0,0,3,3,0 -> rows (0,0,0,2,2), cols (0,0,3,3,0)
    expect = np.ones((4, 4), dtype=np.float32)
    expect[np.arange(4), np.arange(4)] = 0
    expect[:, 0] = 0
    for a, b in zip((0, 0, 0, 2, 2), (0, 0, 3, 3, 0)):
        expect[a, b] = 0
    assert np.array_equal(P, expect)
    assert P[0, 2] == 1.0          # the (0, 2) edge itself is NOT masked: that is the reference's behaviour


def test_fast_sampler_is_stream_identical():
    """Vectorised rejection sampling == the one-by-one loop: same samples, same rejected samples, same check
    count, same state of the global numpy generator afterwards (so later problems see the same stream)."""
    r = _load(golden_files('planner_mazehard_2')[0])
    for n in (1, 37, 500):
        a, b = _env(r), _env(r)
        np.random.seed(99)
        fa, ra = a.sample_n_points(n, need_negative=True)
        nxt_a = np.random.uniform()
        np.random.seed(99)
        fb, rb = b.sample_n_points_fast(n, need_negative=True)
        nxt_b = np.random.uniform()
        assert len(fa) == len(fb) == n and len(ra) == len(rb)
        assert all(np.array_equal(x, y) for x, y in zip(fa, fb))
        assert all(np.array_equal(x, y) for x, y in zip(ra, rb))
        assert a.collision_check_count == b.collision_check_count
        assert nxt_a == nxt_b


def test_attempt_stream_equals_one_by_one_sampling():
    """The shared look-ahead sampler hands consecutive problems exactly the samples (free and rejected), check
    counts and final global-RNG state of the reference's one-by-one rejection loop."""
    from gnnmp.maze2d import AttemptStream, Maze2D
    rng = np.random.RandomState(3)
    maps = (rng.rand(4, 15, 15) < 0.35).astype(np.float64)
    maps[:, 7, 7] = 0
    z = np.zeros((4, 2))
    ref = []
    np.random.seed(99)
    for i in range(4):
        env = Maze2D(maps, z, z)
        env.init_new_problem(i)
        free, rej = env.sample_n_points(40 + 10 * i, need_negative=True)
        ref.append((np.array(free), np.array(rej).reshape(-1, 2), env.collision_check_count))
    tail_ref = np.random.uniform(size=3)
    np.random.seed(99)
    st = AttemptStream(block=64)                    # small block: forces refills in the middle of a problem
    for i in range(4):
        env = Maze2D(maps, z, z)
        env.init_new_problem(i)
        free, rej = env.sample_n_points_stream(st, 40 + 10 * i)
        assert np.array_equal(free, ref[i][0]) and np.array_equal(rej, ref[i][1])
        assert env.collision_check_count == ref[i][2]
    st.close()
    assert np.array_equal(np.random.uniform(size=3), tail_ref)


def test_skip_maze_sampling_reaches_the_sequential_stream_position():
    """Sharded evaluation: skipping the sampling of problems [0, lo) leaves the global RNG where the sequential
    planner loop would be when it starts problem lo."""
    from gnnmp.maze2d import Maze2D
    from gnnmp import planner
    rng = np.random.RandomState(5)
    maps = (rng.rand(6, 15, 15) < 0.3).astype(np.float64)
    maps[:, 0, 0] = 0
    z = np.zeros((6, 2))
    env = Maze2D(maps, z, z)
    np.random.seed(1234)
    for i in range(4):                               # the one-by-one loop through problems 0..3
        e = Maze2D(maps, z, z)
        e.init_new_problem(i)
        e.sample_n_points(30, need_negative=True)
    expect = np.random.uniform(size=4)
    np.random.seed(1234)
    planner.skip_maze_sampling(env, range(4), batch=30)
    assert np.array_equal(np.random.uniform(size=4), expect)


class _StubEnv:
    """Deterministic toy environment for frontier tests: an edge is free unless its (unordered) pair hashes into
    the blocked set; the goal region is one node."""

    def __init__(self, v, blocked, goal_node):
        self.v, self.blocked, self.goal_node = v, blocked, goal_node
        self.collision_check_count = 0

    def _key(self, x):
        return int(np.argmin(np.abs(self.v - x).sum(axis=1)))

    def _edge_fp(self, a, b):
        self.collision_check_count += 1
        i, j = self._key(a), self._key(b)
        return (min(i, j), max(i, j)) not in self.blocked

    def in_goal_region(self, x):
        return self._key(x) == self.goal_node


@pytest.mark.parametrize('seed', range(25))
def test_sparse_frontier_equals_dense_on_random_ties(seed):
    """Heap-based sparse frontier vs the reference's dense masked argmax on random graphs whose scores take only a
    few distinct values (ties everywhere), contain exact zeros, collided nodes, blocked edges and a non-trivial
    starting state (several explored nodes, an explored_edges history that goes through the legacy-index quirk)."""
    from gnnmp import planner
    rng = np.random.RandomState(seed)
    n = int(rng.randint(6, 40))
    v = rng.rand(n, 2).astype(np.float32)
    dense_mask = rng.rand(n, n) < 0.3
    tgt, src = np.nonzero(dense_mask)
    ei = np.stack((src, tgt))                                       # column e: edge src -> tgt = cell P[tgt, src]
    # values whose float32 sums do not cancel to exactly 0: the dense loop of the reference ALSO stops when the sum
    # of the live frontier cells is exactly zero (eval_gnn.py:204), which the sparse frontier does not imitate
    scores = rng.choice(np.array([-2.1, -1.3, 0.0, 0.5, 0.5, 1.7], dtype=np.float32), size=ei.shape[1])
    labels = np.zeros((n, 2), dtype=np.int64)
    labels[rng.rand(n) < 0.15, 1] = 1
    labels[0, 1] = 0
    blocked = set()
    for _ in range(int(rng.randint(0, n))):
        i, j = sorted(rng.randint(0, n, 2).tolist())
        blocked.add((i, j))
    goal = int(rng.randint(1, n))
    start_explored = [0] + [int(x) for x in rng.permutation(np.arange(1, n))[:int(rng.randint(0, 3))]]
    hist = [[0, 0]]
    for _ in range(int(rng.randint(0, 3))):
        a, b = rng.randint(0, n, 2).tolist()
        hist.extend([[a, b], [b, a]])

    def fresh():
        return {'explored': list(start_explored), 'explored_edges': [list(x) for x in hist],
                'costs': {k: 0. for k in start_explored}, 'prev': {k: 0 for k in start_explored}}
    P = np.zeros((n, n), dtype=np.float32)
    P[ei[1], ei[0]] = scores
    sd, ss = fresh(), fresh()
    env_d, env_s = _StubEnv(v, blocked, goal), _StubEnv(v, blocked, goal)
    Pm = planner._mask_policy(P.copy(), labels, sd['explored'], sd['explored_edges'])
    path_d = planner.greedy_expand(Pm, v, env_d, sd)
    path_s = planner.greedy_expand_sparse(scores, ei, labels, v, env_s, ss)
    assert path_d == path_s
    assert sd['explored'] == ss['explored']
    assert sd['explored_edges'] == ss['explored_edges']
    assert env_d.collision_check_count == env_s.collision_check_count


def test_vectorised_host_helpers_equal_the_loops():
    """The device planner's host side replaces per-path Python loops by array passes: the path cost equals
    planner.path_cost (eval_gnn.py:53-58) bit for bit on float32 / float64 waypoint arrays, and the joined chain edge
    lists equal chain_edge_index (smoother.py:238-241) path by path."""
    rng = np.random.default_rng(5)
    for trial in range(400):
        n = int(rng.integers(0, 40))
        dim = 2 + (trial // 2) % 2                         # 2-D point robot rows and 3-D stick robot rows (maze3)
        p = (rng.random((n, dim)) - 0.5).astype(np.float32 if trial % 2 else np.float64)
        assert planner._path_cost_rows(p if n else []) == planner.path_cost(p if n else [])
    lengths = [2, 3, 17, 1, 40, 5]
    joined = planner._chain_edge_indices(lengths)
    ref = torch.cat([planner.chain_edge_index(n) for n in lengths], dim=1).numpy()
    assert joined.dtype == np.int64 and np.array_equal(joined, ref)


def test_graph_batch_takes_strided_tensors():
    """np.argwhere output is column-major (Maze2D.obstacles), and the library reads raw pointers as dense row-major arrays:
    GraphBatch copies strided views into dense tensors (an obstacle array handed over column-major changed 156 of 300 maze
    problems before this was pinned)."""
    from gnnmp.batch import GraphBatch
    obstacles = torch.arange(20, dtype=torch.float32).reshape(2, 10).t()          # [10, 2] with strides (1, 10)
    v = torch.arange(12, dtype=torch.float32).reshape(2, 6).t()
    ei = torch.tensor([[0, 1], [1, 2], [2, 0]]).t()
    assert not obstacles.is_contiguous() and not ei.is_contiguous()
    b = GraphBatch(v, torch.zeros(1, 2), obstacles, ei, torch.tensor([0, 6], dtype=torch.int32),
                   torch.tensor([0, 3], dtype=torch.int32), torch.tensor([0, 10], dtype=torch.int32), 10)
    for t, src in ((b.obstacles, obstacles), (b.v, v), (b.edge_index, ei)):
        assert t.is_contiguous() and torch.equal(t, src)
    env = Maze2D(np.array([[[0, 1, 1], [0, 0, 1], [1, 0, 0]]], dtype=np.float64), np.zeros((1, 2)), np.zeros((1, 2)))
    env.init_new_problem(0)
    joined = np.ascontiguousarray(np.concatenate([np.asarray(env.obstacles).reshape(-1, 2)] * 2), dtype=np.float32)
    assert torch.from_numpy(joined).is_contiguous()


@pytest.mark.parametrize('n,chunk', [(0, 128), (1, 128), (127, 128), (128, 128), (129, 128), (1000, 512), (1024, 128), (77, 1), (5000, 3)])
def test_device_pass_spans_partition_the_problems(n, chunk):
    """planner.eval_gnn_device cuts an evaluation into device passes (a short ramp, then passes of `chunk`): whatever the sizes,
    the spans cover range(n) exactly once and in order -- the sampler thread consumes the global numpy stream in that order."""
    spans = planner._pass_spans(n, chunk)
    flat = [i for lo, hi in spans for i in range(lo, hi)]
    assert flat == list(range(n))
    assert all(0 < hi - lo <= chunk for lo, hi in spans)
    if n > chunk >= 4:
        assert spans[0][1] - spans[0][0] == chunk // 4
        if n - chunk // 4 > chunk:
            assert spans[1][1] - spans[1][0] == chunk // 2

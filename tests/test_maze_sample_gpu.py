"""Rejection sampling of the explore stage on the device (gnnmp_maze_sample, csrc/maze_kernels.hip) against the host sampler.

The reference draws a problem's nodes one by one from numpy's GLOBAL generator (eval_gnn.py:180-184 -> MazeEnv.sample_n_points,
environment/maze_env.py) and the next problem continues in the same stream.  gnnmp.planner.sample_maze_problems is the vectorised
host counterpart (identical samples / counts / generator state, tests/test_planner_host.py); sample_maze_problems_device keeps only
the draws on the host.  Asserted: the same float32 node rows, the same per-problem check counts, the same generator state
afterwards -- also when the first block of draws handed to the device is too short and the launch has to be repeated."""
import ctypes
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from gnnmp import _lib, planner

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _problems(n):
    with np.load(os.path.join(GOLDEN, 'evalset_mazehard_first1000.npz')) as f:
        maps, init, goal = f['maps'], f['init_states'], f['goal_states']
    return [dict(map=maps[i], init_state=init[i], goal_state=goal[i]) for i in range(n)]


def _same_state(a, b):
    return a[0] == b[0] and np.array_equal(a[1], b[1]) and a[2:] == b[2:]


@pytest.mark.parametrize('batch,estimate', [(500, 3.0), (200, 0.4), (37, 3.0)], ids=['published_setting', 'short_first_block', 'odd_count'])
def test_device_sampler_equals_host_sampler(batch, estimate):
    pr = _problems(48)
    np.random.seed(11)
    envs, vs, n_free, k1s = planner.sample_maze_problems(pr, batch, 30)
    st_host = np.random.get_state()
    np.random.seed(11)
    planner._DRAWS_PER_FREE[0] = estimate                      # 0.4: the first block cannot hold the draws -> the launch is repeated
    d = planner.sample_maze_problems_device(pr, batch, 30, DEV)
    st_dev = np.random.get_state()
    assert _same_state(st_host, st_dev)                        # the global generator is where one-by-one sampling leaves it
    nptr = d['node_ptr_host']
    v = d['v'].cpu()
    assert int(nptr[-1]) == sum(x.shape[0] for x in vs) == v.shape[0]
    for b in range(len(pr)):
        assert torch.equal(v[nptr[b]:nptr[b + 1]], vs[b]), b   # [start, goal, free ..., rejected[:batch] ...], float32, bit for bit
        assert d['envs'][b].collision_check_count == envs[b].collision_check_count
    assert d['n_free'] == n_free and d['k1s'] == k1s
    assert d['node_ptr'].cpu().tolist() == [int(x) for x in nptr]


@pytest.mark.parametrize('w', [15, 70], ids=['map_in_lds', 'map_in_global_memory'])
def test_raw_abi_against_a_numpy_restatement(w):
    """gnnmp_maze_sample through the C ABI on synthetic maps (15 x 15 staged in LDS, 70 x 70 read from global memory) against a
    direct numpy restatement of the rule: cell = ((p + 1) * w / 2).astype(int) clipped at w - 1, free <=> map[cell] == 0."""
    rng = np.random.default_rng(3)
    B, n = 9, 64
    maps = (rng.random((B, w, w)) < 0.45).astype(np.float64)
    maps[:, 0, 0] = 0.0
    init = rng.uniform(-1, 1, (B, 2)); goal = rng.uniform(-1, 1, (B, 2))
    att = rng.uniform(-1, 1, (B * n * 6, 2))
    rows, ptr, used, cur = [], [0], [], 0
    for b in range(B):
        free, rej = [], []
        while len(free) < n:
            p = att[cur]; cur += 1
            c = ((p + 1.0) * w / 2.0).astype(int); c[c > w - 1] = w - 1
            (free if maps[b][c[0], c[1]] == 0 else rej).append(p)
        used.append(len(free) + len(rej))
        rows.append(np.concatenate((init[b:b + 1], goal[b:b + 1], np.array(free), np.array(rej[:n]).reshape(-1, 2))).astype(np.float32))
        ptr.append(ptr[-1] + rows[-1].shape[0])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)      # noqa: E731
    d_att, d_maps, d_init, d_goal = t(att), t(maps), t(init), t(goal)
    v = torch.full((B * (2 + 2 * n), 2), float('nan'), dtype=torch.float32, device=DEV)
    nptr = torch.zeros(B + 1, dtype=torch.int32, device=DEV); d_used = torch.zeros(B, dtype=torch.int32, device=DEV)
    state = torch.zeros(2, dtype=torch.int64, device=DEV)
    sb = _lib.MazeSampleBatch(B, w, n, att.shape[0], d_att.data_ptr(), d_maps.data_ptr(), d_init.data_ptr(), d_goal.data_ptr())
    rc = _lib.lib().gnnmp_maze_sample(ctypes.byref(sb), state.data_ptr(), v.data_ptr(), nptr.data_ptr(), d_used.data_ptr(),
                                      state.data_ptr() + 8, None)
    assert rc == 0
    torch.cuda.synchronize()
    cursor, ok = state.cpu().tolist()
    assert ok == 1 and cursor == cur
    assert nptr.cpu().tolist() == ptr and d_used.cpu().tolist() == used
    assert np.array_equal(v[:ptr[-1]].cpu().numpy(), np.concatenate(rows))
    # a stream that ends inside the last problem: ok = 0, nothing consumed
    sb.n_attempts = cur - 3
    state.zero_()
    assert _lib.lib().gnnmp_maze_sample(ctypes.byref(sb), state.data_ptr(), v.data_ptr(), nptr.data_ptr(), d_used.data_ptr(),
                                        state.data_ptr() + 8, None) == 0
    torch.cuda.synchronize()
    assert state.cpu().tolist() == [0, 0]
    # NULL arguments are refused
    assert _lib.lib().gnnmp_maze_sample(ctypes.byref(sb), None, v.data_ptr(), nptr.data_ptr(), d_used.data_ptr(), state.data_ptr() + 8, None) == -1

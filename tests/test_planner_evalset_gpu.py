"""End-to-end known-answer at the reference's DEFAULT planner configuration (eval_gnn defaults batch=500,
t_max=500, k=30 -> N ~ 1002 nodes, k1 = 41, E ~ 56 k, smoothing on; mazes_hard.npz, seed 1234 -- the setting
of the notebook's published run, main.ipynb:57-61): the GPU planner must reproduce, problem by problem, the
outcomes the reference planner produced with its own CPU models (tests/golden/evalset_*.npz), in both the
drop-in dense mode and the sparse-frontier + device-graph mode."""
import numpy as np
import pytest
import torch

from conftest import golden_files, load_weights
import gnnmp
from gnnmp import planner
from gnnmp.maze2d import Maze2D

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.mark.parametrize('sparse,gpu_graph', [(False, False), (True, False), (True, True)],
                         ids=['dense_dropin', 'sparse_frontier', 'sparse_frontier_device_graph'])
def test_eval_set_matches_reference(sparse, gpu_graph):
    with np.load(golden_files('evalset_mazehard_first12')[0]) as f:
        r = {k: f[k] for k in f.files}
    env = Maze2D(r['maps'], r['init_states'], r['goal_states'])
    m = gnnmp.EncoderProcessDecoder(2, 2, 32, 2).eval()
    m.load_state_dict(load_weights('weights_maze'))
    ms = gnnmp.ModelSmoother(workspace_size=2, config_size=2, embed_size=128, obs_size=6).eval()
    ms.load_state_dict(load_weights('smooth_2d_attv3'))
    np.random.seed(int(r['seed']))
    torch.manual_seed(int(r['seed']))
    rows = []
    for idx in range(r['rows'].shape[0]):
        env.init_new_problem(idx)
        res = planner.explore(env, m, ms, True, batch=int(r['batch']), t_max=int(r['t_max']), k=int(r['k']), device=DEV,
                              sparse=sparse, gpu_graph=gpu_graph)
        rows.append([int(res['success']), planner.path_cost(res['path']), planner.path_cost(res['smooth_path']),
                     res['c_explore'], res['c_smooth'], len(res['path']), len(res['explored'])])
    rows = np.array(rows, dtype=np.float64)
    ref = r['rows']
    print('\nc_explore  gpu %s\n           ref %s' % (rows[:, 3].astype(int).tolist(), ref[:, 3].astype(int).tolist()))
    print('c_smooth   gpu %s\n           ref %s' % (rows[:, 4].astype(int).tolist(), ref[:, 4].astype(int).tolist()))
    assert np.array_equal(rows[:, 0], ref[:, 0])                       # success
    assert np.array_equal(rows[:, 3], ref[:, 3])                       # collision checks, explore stage
    assert np.array_equal(rows[:, 6], ref[:, 6])                       # explored nodes
    assert np.allclose(rows[:, 1], ref[:, 1], rtol=0, atol=1e-9)       # raw path cost (same nodes)
    # smoothing: same steering decisions unless a proposal sits within fp32 noise of the RRT_EPS threshold
    assert np.allclose(rows[:, 2], ref[:, 2], rtol=1e-3, atol=1e-3)
    assert np.abs(rows[:, 4] - ref[:, 4]).max() <= 0.03 * ref[:, 4].max()

"""End-to-end planner on the GPU (host control flow + HIP models) against the traces recorded from the
reference planner with its own CPU models (tests/golden/planner_*.npz): every forward's scores
within tolerance, same success, and -- since the planner consumes orderings -- the same explored-node
sequence and collision-check count wherever no decision sat inside the fp32 noise floor."""
import os

import numpy as np
import pytest
import torch

from conftest import golden_files, load_weights
import gnnmp
from gnnmp import planner
from gnnmp.maze2d import Maze2D

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _load(path):
    with np.load(path) as f:
        return {k: f[k] for k in f.files}


@pytest.mark.parametrize('path', golden_files('planner_'), ids=os.path.basename)
def test_gpu_planner_reproduces_reference_run(path):
    r = _load(path)
    env = Maze2D(r['map'][None], r['init_state'][None], r['goal_state'][None])
    env.init_new_problem(0)
    m = gnnmp.EncoderProcessDecoder(2, 2, 32, 2).eval()
    m.load_state_dict(load_weights('weights_maze'))
    ms = gnnmp.ModelSmoother(workspace_size=2, config_size=2, embed_size=128, obs_size=6).eval()
    ms.load_state_dict(load_weights('smooth_2d_attv3'))
    np.random.seed(int(r['seed']))
    torch.manual_seed(int(r['seed']))
    trace = {}
    res = planner.explore(env, m, ms, True, batch=int(r['batch']), t_max=int(r['t_max']), k=int(r['k']), device=DEV,
                          trace=trace)
    assert res['success'] == bool(r['success'])
    # first forward sees exactly the reference's inputs: scores must match within the fp32 bar
    f0 = trace['forwards'][0]
    assert np.array_equal(f0['edge_index'], r['e0_edge_index'])
    # (trace fixtures hold the reference's fp32 scores only; the N ~ 1000, O ~ 116 maze noise floor is 2.4e-5, see
    # profiles/r02_parity.txt -- the goldens with an fp64 run carry the tight per-fixture bar)
    assert np.allclose(f0['scores'], r['e0_scores'], rtol=1e-5, atol=3e-5)
    same_walk = res['explored'] == r['explored'].tolist()
    print('\n%s: identical explored sequence: %s; c_explore %d vs %d; c_smooth %d vs %d' %
          (os.path.basename(path), same_walk, res['c_explore'], int(r['c_explore']), res['c_smooth'], int(r['c_smooth'])))
    assert same_walk
    assert res['explored_edges'] == r['explored_edges'].tolist()
    assert res['c_explore'] == int(r['c_explore'])
    assert np.array_equal(np.array(res['path'], dtype=np.float32), r['path'])
    # smoothing: network proposals within tolerance; steering decisions identical -> same final path
    for i, (p_in, p_out) in enumerate(trace['smooth']):
        if np.array_equal(p_in, r['s%d_path' % i]):
            assert np.allclose(p_out, r['s%d_out' % i], rtol=1e-5, atol=1e-5)
    assert np.allclose(np.array(res['smooth_path'], dtype=np.float64), r['smooth_path'], rtol=1e-4, atol=1e-4)
    assert abs(res['c_smooth'] - int(r['c_smooth'])) <= max(8, int(0.02 * int(r['c_smooth'])))

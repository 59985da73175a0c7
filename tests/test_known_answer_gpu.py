"""The reference's only published known-answer (main.ipynb:50-63: eval_gnn on the 1000 problems of mazes_hard.npz,
seed 1234, batch = t_max = 500, k = 30, smoothing on) through the device planner (planner.eval_gnn_device: graphs,
explorer forward, greedy loop + collision checks, smoothing -- all on the GPU):

 (a) against the per-problem outcomes of the unmodified reference planner run on the CPU in the authoring
     container (tests/golden/evalset_mazehard_first1000.npz, generator: tools/gen_golden.py evalset 1000), and
 (b) against the aggregates printed in the notebook (the explore stage is reproduced by the
     shipped checkpoint, the smoothing stage of that run is not, so only the explore-stage numbers are asserted)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_weights
import gnnmp
from gnnmp import planner
from gnnmp.maze2d import Maze2D

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
PUBLISHED = {'success': 1000, 'collision_total': 2487.37, 'collision_explore': 2212.69, 'path_cost': 2.19}   # main.ipynb:57-61


def test_published_run_1000_problems():
    path = os.path.join(GOLDEN, 'evalset_mazehard_first1000.npz')
    with np.load(path) as f:
        r = {k: f[k] for k in f.files}
    env = Maze2D(r['maps'], r['init_states'], r['goal_states'])
    m = gnnmp.EncoderProcessDecoder(2, 2, 32, 2).eval()
    m.load_state_dict(load_weights('weights_maze'))
    ms = gnnmp.ModelSmoother(workspace_size=2, config_size=2, embed_size=128, obs_size=6).eval()
    ms.load_state_dict(load_weights('smooth_2d_attv3'))
    rows = []
    out = planner.eval_gnn_device(env, range(1000), m, ms, seed=int(r['seed']), batch=int(r['batch']), k=int(r['k']),
                                  device=DEV, rows_out=rows)
    n_success, collision, _, cost, total_time, _, _, collision_explore, _ = out
    rows = np.array(rows, dtype=np.float64)
    ref = r['rows']
    print('\ndevice planner: success %d, checks %.2f (explore %.2f), smoothed cost %.4f, %.2f s for 1000 problems'
          % (n_success, collision, collision_explore, cost, total_time))
    print('reference (CPU, this container): success %d, checks %.2f (explore %.2f), smoothed cost %.4f'
          % (ref[:, 0].sum(), (ref[:, 3] + ref[:, 4]).mean(), ref[:, 3].mean(), ref[ref[:, 0] > 0, 2].mean()))
    print('published (main.ipynb:57-61):', PUBLISHED)
    # (a) per problem against the reference's own run.  Measured (profiles/r03_planner_parity.txt): 1000 / 1000 identical in
    # both stages.  The asserts keep a margin of 2 / 5 problems: the GPU forward differs from the CPU forward by up to 2e-5
    # on the edge scores, so a later change of kernel rounding may resolve a near-tie between two frontier edges differently.
    same = (rows[:, 3] == ref[:, 3]) & (rows[:, 6] == ref[:, 6]) & (rows[:, 5] == ref[:, 5])
    print('explore stage identical (checks, explored nodes, path length) on %d / 1000 problems' % int(same.sum()))
    assert np.array_equal(rows[:, 0], ref[:, 0])
    assert same.sum() >= 998
    assert abs(rows[:, 3].mean() - ref[:, 3].mean()) <= 0.002 * ref[:, 3].mean()
    sm_same = same & (rows[:, 4] == ref[:, 4])
    print('smoothing stage identical check counts on %d of those' % int(sm_same.sum()))
    assert sm_same.sum() >= 995
    assert abs(cost - ref[ref[:, 0] > 0, 2].mean()) <= 2e-3
    # (b) the notebook
    assert n_success == PUBLISHED['success']
    assert abs(collision_explore - PUBLISHED['collision_explore']) <= 0.001 * PUBLISHED['collision_explore']


def test_published_run_in_bf16_mode_decision_level():
    """Decision-level acceptance of the bf16 operand mode (BASELINE configs[2] / [4] run the explorer with bf16 MFMA operands).
    The consumer of the scores is an argmax over frontier edges (eval_gnn.py:205-211), so what the mode has to preserve is the
    planner's OUTCOME: the same 1000 problems and samples as the published run (the sampling stream does not depend on the
    scores), explorer in bf16 mode.  Measured (profiles/r05_planner_parity.txt): success 1000 / 1000, mean explore checks
    2213.83 against 2212.49 in fp32 (+0.06 %; published 2212.69), smoothed cost 2.2325 against 2.2308 (+0.08 %); the explore
    stage is step-for-step identical on 375 problems (a near-tie between two frontier edges resolved differently changes the
    rest of that problem's search, mean difference +1.3 checks, median 0).  Asserted: every problem solved, both aggregates
    within 0.5 % of the fp32 run."""
    path = os.path.join(GOLDEN, 'evalset_mazehard_first1000.npz')
    with np.load(path) as f:
        r = {k: f[k] for k in f.files}
    env = Maze2D(r['maps'], r['init_states'], r['goal_states'])
    m = gnnmp.EncoderProcessDecoder(2, 2, 32, 2).eval()
    m.load_state_dict(load_weights('weights_maze'))
    m.mlp_dtype = 'bf16'
    ms = gnnmp.ModelSmoother(workspace_size=2, config_size=2, embed_size=128, obs_size=6).eval()
    ms.load_state_dict(load_weights('smooth_2d_attv3'))
    rows = []
    out = planner.eval_gnn_device(env, range(1000), m, ms, seed=int(r['seed']), batch=int(r['batch']), k=int(r['k']),
                                  device=DEV, rows_out=rows)
    rows = np.array(rows, dtype=np.float64)
    ref = r['rows']
    same = (rows[:, 3] == ref[:, 3]) & (rows[:, 6] == ref[:, 6]) & (rows[:, 5] == ref[:, 5])
    print('\nbf16 explorer: success %d, mean explore checks %.2f (fp32 %.2f), smoothed cost %.4f (fp32 %.4f), identical explore stage on %d'
          % (out[0], rows[:, 3].mean(), ref[:, 3].mean(), out[3], ref[ref[:, 0] > 0, 2].mean(), int(same.sum())))
    assert out[0] == 1000 and np.array_equal(rows[:, 0], ref[:, 0])
    assert abs(rows[:, 3].mean() / ref[:, 3].mean() - 1.0) <= 0.005
    assert abs(out[3] / ref[ref[:, 0] > 0, 2].mean() - 1.0) <= 0.005
    assert same.sum() >= 250                                   # measured 375: most problems have at least one near-tie somewhere


def test_sharded_evaluation_equals_sequential():
    """Two shards (rank 0 / rank 1 of 2, run one after the other here) of the first 64 problems: the union of the
    per-problem rows equals the single-process run (the skipped sampling puts each shard at the right RNG position)."""
    path = os.path.join(GOLDEN, 'evalset_mazehard_first1000.npz')
    with np.load(path) as f:
        r = {k: f[k] for k in f.files}
    env = Maze2D(r['maps'], r['init_states'], r['goal_states'])
    m = gnnmp.EncoderProcessDecoder(2, 2, 32, 2).eval()
    m.load_state_dict(load_weights('weights_maze'))
    ms = gnnmp.ModelSmoother(workspace_size=2, config_size=2, embed_size=128, obs_size=6).eval()
    ms.load_state_dict(load_weights('smooth_2d_attv3'))
    whole, parts = [], []
    planner.eval_gnn_device(env, range(64), m, ms, device=DEV, rows_out=whole)
    for rank in range(2):
        planner.eval_gnn_device(env, range(64), m, ms, device=DEV, rows_out=parts, shard=(rank, 2))
    assert len(parts) == 64
    assert np.array_equal(np.array(whole), np.array(parts))


def test_worker_threads_do_not_change_any_problem():
    """Device passes of several chunks in flight on their own streams (``workers`` host threads): the per-problem rows of
    192 problems in chunks of 32 are the same with 1, 2 and 3 workers (chunks are independent, the sampling stays on one
    thread in problem order, and each stream has its own workspaces)."""
    with np.load(os.path.join(GOLDEN, 'evalset_mazehard_first1000.npz')) as f:
        env = Maze2D(f['maps'], f['init_states'], f['goal_states'])
    m = gnnmp.EncoderProcessDecoder(2, 2, 32, 2).eval()
    m.load_state_dict(load_weights('weights_maze'))
    ms = gnnmp.ModelSmoother(workspace_size=2, config_size=2, embed_size=128, obs_size=6).eval()
    ms.load_state_dict(load_weights('smooth_2d_attv3'))
    rows = {}
    for workers in (1, 2, 3):
        rows[workers] = []
        out = planner.eval_gnn_device(env, range(192), m, ms, device=DEV, chunk=32, workers=workers, rows_out=rows[workers])
        assert out[0] == sum(r[0] for r in rows[workers])
    assert np.array_equal(np.array(rows[1]), np.array(rows[2]))
    assert np.array_equal(np.array(rows[1]), np.array(rows[3]))


def test_second_setting_400_problems():
    """The same problem set under another planner setting (batch = t_max = 200, k = 16, seed 7; smaller graphs, some
    problems unsolved in one round): per-problem outcomes of the unmodified reference (tools/gen_golden.py evalset 400
    200 16 7) against the device planner."""
    with np.load(os.path.join(GOLDEN, 'evalset_mazehard_first1000.npz')) as f:
        env = Maze2D(f['maps'], f['init_states'], f['goal_states'])
    with np.load(os.path.join(GOLDEN, 'evalrows_mazehard_first400_b200_k16_s7.npz')) as f:
        ref, seed, batch, k = f['rows'], int(f['seed']), int(f['batch']), int(f['k'])
    m = gnnmp.EncoderProcessDecoder(2, 2, 32, 2).eval()
    m.load_state_dict(load_weights('weights_maze'))
    ms = gnnmp.ModelSmoother(workspace_size=2, config_size=2, embed_size=128, obs_size=6).eval()
    ms.load_state_dict(load_weights('smooth_2d_attv3'))
    rows = []
    planner.eval_gnn_device(env, range(ref.shape[0]), m, ms, seed=seed, batch=batch, k=k, device=DEV, rows_out=rows)
    rows = np.array(rows, dtype=np.float64)
    same = (rows[:, 0] == ref[:, 0]) & (rows[:, 3] == ref[:, 3]) & (rows[:, 6] == ref[:, 6]) & (rows[:, 5] == ref[:, 5])
    sm_same = same & (rows[:, 4] == ref[:, 4])
    print('\nsolved %d (reference %d) of %d; explore stage identical on %d, smoothing check counts on %d'
          % (rows[:, 0].sum(), ref[:, 0].sum(), ref.shape[0], same.sum(), sm_same.sum()))
    assert np.array_equal(rows[:, 0], ref[:, 0])
    assert same.sum() >= 398                                  # measured 400 / 400 (profiles/r03_planner_parity.txt); margin 2
    assert sm_same.sum() >= 396                               # measured 400

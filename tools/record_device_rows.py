#!/usr/bin/env python
"""Run ON THE GPU BOX: per-problem result rows of planner.eval_gnn_device (device planner) for the first 64 problems of
the published run's setting -> gpurun_out/evalrows_device_first64.npz (copied to tests/golden/ as the fixture the
world-size-2 gloo test shards and gathers, tests/test_dist_gloo.py)."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import gnnmp  # noqa: E402
from gnnmp import planner  # noqa: E402
from gnnmp.maze2d import Maze2D  # noqa: E402
from gnnmp.weights import load_weights  # noqa: E402

with np.load(os.path.join(REPO, 'tests', 'golden', 'evalset_mazehard_first1000.npz')) as f:
    r = {k: f[k] for k in f.files}
env = Maze2D(r['maps'], r['init_states'], r['goal_states'])
m = gnnmp.EncoderProcessDecoder(2, 2, 32, 2).eval()
m.load_state_dict(load_weights('weights_maze'))
ms = gnnmp.ModelSmoother(workspace_size=2, config_size=2, embed_size=128, obs_size=6).eval()
ms.load_state_dict(load_weights('smooth_2d_attv3'))
rows = []
planner.eval_gnn_device(env, range(64), m, ms, seed=int(r['seed']), batch=int(r['batch']), k=int(r['k']), device='cuda:0',
                        rows_out=rows)
rows = np.array(rows, dtype=np.float64)
os.makedirs(os.path.join(REPO, 'gpurun_out'), exist_ok=True)
np.savez(os.path.join(REPO, 'gpurun_out', 'evalrows_device_first64.npz'), rows=rows, seed=r['seed'], batch=r['batch'], k=r['k'])
print(rows.shape, rows[:3])

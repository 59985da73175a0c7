#!/usr/bin/env python
"""eval_gnn_device at 1024 problems with host sampling vs device sampling (gnnmp_maze_sample), workers 1 / 2; median of 3."""
import os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'tests'))
import gnnmp
from gnnmp import planner
from gnnmp.maze2d import Maze2D
from gnnmp.weights import load_weights
with np.load(os.path.join(REPO, 'tests', 'golden', 'evalset_mazehard_first1000.npz')) as f:
    env = Maze2D(f['maps'], f['init_states'], f['goal_states'])
m = gnnmp.EncoderProcessDecoder(2, 2, 32, 2).eval(); m.load_state_dict(load_weights('weights_maze'))
ms = gnnmp.ModelSmoother(workspace_size=2, config_size=2, embed_size=128, obs_size=6).eval(); ms.load_state_dict(load_weights('smooth_2d_attv3'))
idx = [i % 1000 for i in range(1024)]
ref = None
for workers in (2, 1):
    for dsamp in (False, True):
        rows = []
        planner.eval_gnn_device(env, idx, m, ms, device='cuda:0', workers=workers, device_sampling=dsamp)
        ts = []
        for _ in range(3):
            rows = []
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = planner.eval_gnn_device(env, idx, m, ms, device='cuda:0', workers=workers, device_sampling=dsamp, rows_out=rows)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        rows = np.array(rows)
        if ref is None: ref = rows
        print('workers %d device_sampling %-5s: %s problems/s (median %.0f); success %d checks %.2f; rows identical to the first variant: %s' % (
            workers, dsamp, ' / '.join('%.0f' % (1024 / t) for t in ts), 1024 / sorted(ts)[1], out[0], out[1], bool(np.array_equal(rows, ref))), flush=True)

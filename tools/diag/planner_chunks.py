#!/usr/bin/env python
"""Run ON THE GPU BOX: device planner rate (planner.eval_gnn_device, 1024 problems of the published run's setting, smoothing on)
against the chunk size and the number of device-pass worker threads (the sampling runs ahead on its own host thread)."""
import os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import gnnmp
from gnnmp import planner
from gnnmp.maze2d import Maze2D
from gnnmp.weights import load_weights
dev = 'cuda:0'
with np.load(os.path.join(REPO, 'tests', 'golden', 'evalset_mazehard_first1000.npz')) as f:
    env = Maze2D(f['maps'], f['init_states'], f['goal_states'])
m = gnnmp.EncoderProcessDecoder(2, 2, 32, 2).eval(); m.load_state_dict(load_weights('weights_maze'))
ms = gnnmp.ModelSmoother(workspace_size=2, config_size=2, embed_size=128, obs_size=6).eval(); ms.load_state_dict(load_weights('smooth_2d_attv3'))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
idx = [i % len(env.maps) for i in range(n)]
for workers in (1, 2, 3):
    for chunk in (512, 256, 128):
        planner.eval_gnn_device(env, idx, m, ms, device=dev, chunk=chunk, workers=workers)
        walls = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = planner.eval_gnn_device(env, idx, m, ms, device=dev, chunk=chunk, workers=workers)
            torch.cuda.synchronize()
            walls.append(time.perf_counter() - t0)
        print('workers %d chunk %4d: %s problems/s (median %.0f), success %d, checks %.2f, reserved %.1f GB' % (workers, chunk, ' / '.join('%.0f' % (n / w) for w in walls), n / sorted(walls)[1], out[0], out[1], torch.cuda.memory_reserved() / 2**30), flush=True)

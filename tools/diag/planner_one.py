#!/usr/bin/env python
"""Run ON THE GPU BOX: device planner, one configuration (problems, chunk, workers), 1 warm-up + 3 timed runs."""
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
n, chunk, workers = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
idx = [i % len(env.maps) for i in range(n)]
planner.eval_gnn_device(env, idx, m, ms, device=dev, chunk=chunk, workers=workers)
walls = []
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = planner.eval_gnn_device(env, idx, m, ms, device=dev, chunk=chunk, workers=workers)
    torch.cuda.synchronize(); walls.append(time.perf_counter() - t0)
print('problems %d chunk %d workers %d: %s problems/s, checks %.2f' % (n, chunk, workers, ' / '.join('%.0f' % (n / w) for w in walls), out[1]))

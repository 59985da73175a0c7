#!/usr/bin/env python
"""Run ON THE GPU BOX: cProfile of the host side of one device-planner evaluation (main thread = device passes; the
sampler thread is profiled separately with --sampler, which needs no GPU)."""
import cProfile, os, pstats, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import gnnmp
from gnnmp import planner
from gnnmp.maze2d import Maze2D
from gnnmp.weights import load_weights
with np.load(os.path.join(REPO, 'tests', 'golden', 'evalset_mazehard_first1000.npz')) as f:
    env = Maze2D(f['maps'], f['init_states'], f['goal_states'])
n = 1024
idx = [i % len(env.maps) for i in range(n)]
if '--sampler' in sys.argv:
    pr = [dict(map=env.maps[i], init_state=env.init_states[i], goal_state=env.goal_states[i]) for i in idx[:512]]
    np.random.seed(1)
    planner.sample_maze_problems(pr, 500, 30)
    t0 = time.perf_counter()
    prof = cProfile.Profile(); prof.enable()
    planner.sample_maze_problems(pr, 500, 30)
    prof.disable()
    print('sampling of 512 problems: %.1f ms' % (1e3 * (time.perf_counter() - t0)))
    pstats.Stats(prof).sort_stats('tottime').print_stats(18)
    sys.exit(0)
dev = 'cuda:0'
m = gnnmp.EncoderProcessDecoder(2, 2, 32, 2).eval(); m.load_state_dict(load_weights('weights_maze'))
ms = gnnmp.ModelSmoother(workspace_size=2, config_size=2, embed_size=128, obs_size=6).eval(); ms.load_state_dict(load_weights('smooth_2d_attv3'))
planner.eval_gnn_device(env, idx, m, ms, device=dev, chunk=256, workers=1)
prof = cProfile.Profile(); prof.enable()
t0 = time.perf_counter()
planner.eval_gnn_device(env, idx, m, ms, device=dev, chunk=256, workers=1)
print('wall %.1f ms' % (1e3 * (time.perf_counter() - t0)))
prof.disable()
pstats.Stats(prof).sort_stats('tottime').print_stats(45)

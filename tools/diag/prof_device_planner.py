#!/usr/bin/env python
"""Run ON THE GPU BOX: cProfile of planner.eval_gnn_device over 1024 problems (chunk 512) -- where the host time of the device
planner goes (main thread only; the sampler thread of the next chunk runs beside it)."""
import cProfile, os, pstats, sys, time
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
idx = [i % len(env.maps) for i in range(1024)]
planner.eval_gnn_device(env, idx, m, ms, device=dev, chunk=512)
tm = {}
probs = [dict(map=env.maps[i], init_state=env.init_states[i], goal_state=env.goal_states[i]) for i in idx[:512]]
np.random.seed(1)
t0 = time.perf_counter(); pre = planner.sample_maze_problems(probs, 500, 30); t1 = time.perf_counter()
res = planner.explore_maze_batch(probs, m, dev, batch=500, k=30, model_s=ms, timings=tm, presampled=pre); torch.cuda.synchronize(); t2 = time.perf_counter()
print('512 problems: sampling %.1f ms, device pass %.1f ms (its own stage timers: %s)' % (1e3 * (t1 - t0), 1e3 * (t2 - t1), {k: round(1e3 * v, 1) for k, v in tm.items()}))
pr = cProfile.Profile()
pr.enable()
planner.explore_maze_batch(probs, m, dev, batch=500, k=30, model_s=ms, presampled=pre); torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)

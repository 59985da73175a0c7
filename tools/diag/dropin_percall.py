"""Per-problem split of the drop-in forward span + garbage-collector pauses.  python tools/diag/dropin_percall.py"""
import gc, os, sys, time
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import gnnmp
from gnnmp import planner
from gnnmp.maze2d import Maze2D
from gnnmp.weights import load_weights

dev = torch.device('cuda:0')
with np.load(os.path.join(REPO, 'tests', 'golden', 'evalset_mazehard_first1000.npz')) as f:
    env = Maze2D(f['maps'], f['init_states'], f['goal_states'])
m = gnnmp.EncoderProcessDecoder(2, 2, 32, 2).eval(); m.load_state_dict(load_weights('weights_maze'))
ms = gnnmp.ModelSmoother(workspace_size=2, config_size=2, embed_size=128, obs_size=6).eval(); ms.load_state_dict(load_weights('smooth_2d_attv3'))
pauses, t_gc = [], [0.]
def cb(phase, info):
    if phase == 'start':
        t_gc[0] = time.perf_counter()
    else:
        pauses.append((info['generation'], (time.perf_counter() - t_gc[0]) * 1e3))
gc.callbacks.append(cb)
np.random.seed(1234)
for rnd in range(3):
    for i in range(16):
        env.init_new_problem(i)
        pauses.clear()
        r = planner.explore(env, m, ms, True, batch=500, t_max=500, k=30, device=dev)
        s = r['forward_split']
        big = [(g, round(p, 2)) for g, p in pauses if p > 0.3]
        print('round %d problem %2d forward %7.3f ms  obs %6.3f h2d %6.3f call %6.3f d2h %6.3f  E %d  gc>0.3ms %s' % (
            rnd, i, 1e3 * r['forward'], 1e3 * s['obs_data'], 1e3 * s['h2d'], 1e3 * s['module_call'], 1e3 * s['d2h_wait'],
            r['data']['edge_index'].shape[1], big), flush=True)
    if rnd == 0:
        torch.cuda.empty_cache()
print('objects tracked by gc:', len(gc.get_objects()))

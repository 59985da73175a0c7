#!/usr/bin/env python
"""Run ON THE GPU BOX with a -DGNNMP_MAZE_TRACE build (GNNMP_LIB=.../libgnnmp_mztrace.so): where the cycles of the greedy explore
kernel go -- per problem: CSR build, argmax over the cached row maxima, collision check (lane 0), cell kills, row rescans
(wall_clock64 ticks of 10 ns), for one device pass of 256 problems."""
import ctypes, os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import gnnmp
from gnnmp import planner, _lib
from gnnmp.maze2d import Maze2D
from gnnmp.weights import load_weights
dev = 'cuda:0'
with np.load(os.path.join(REPO, 'tests', 'golden', 'evalset_mazehard_first1000.npz')) as f:
    env = Maze2D(f['maps'], f['init_states'], f['goal_states'])
m = gnnmp.EncoderProcessDecoder(2, 2, 32, 2).eval(); m.load_state_dict(load_weights('weights_maze'))
n = 256
probs = [dict(map=env.maps[i], init_state=env.init_states[i], goal_state=env.goal_states[i]) for i in range(n)]
np.random.seed(1234)
planner.explore_maze_batch(probs, m, dev, batch=500, k=30)
torch.cuda.synchronize()
buf = np.zeros(8 * n, dtype=np.int64)
L = _lib.lib()
assert L.gnnmp_debug_maze_trace(buf.ctypes.data_as(ctypes.c_void_p), n) == 0
t = buf.reshape(n, 8).astype(np.float64)
us = t[:, :5] * 0.01
names = ['build', 'argmax', 'check', 'kill', 'rescan']
tot = us.sum(1)
print('problems %d: mean total %.0f us, max %.0f us; steps mean %.0f max %.0f; explored mean %.0f; edges mean %.0f' % (n, tot.mean(), tot.max(), t[:, 5].mean(), t[:, 5].max(), t[:, 6].mean(), t[:, 7].mean()))
print('mean us per phase   : ' + ', '.join('%s %.0f' % (k, v) for k, v in zip(names, us.mean(0))))
w = int(tot.argmax())
print('slowest problem %d   : ' % w + ', '.join('%s %.0f' % (k, v) for k, v in zip(names, us[w])) + ', steps %d, explored %d' % (t[w, 5], t[w, 6]))
steps = np.maximum(t[:, 5], 1)
print('mean us per step    : ' + ', '.join('%s %.2f' % (k, v) for k, v in zip(names[1:], (us[:, 1:] / steps[:, None]).mean(0))))

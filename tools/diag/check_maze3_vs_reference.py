#!/usr/bin/env python
"""Authoring-container check (needs /root/reference): gnnmp.maze2d.Maze3D against the reference's MazeEnv(dim=3), query by
query, on random float32 configurations: distance, interpolate, edge checks (result and collision-check count), state
checks.  Run after touching maze2d.py."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = '/root/reference'
sys.path.insert(0, os.path.join(REPO, 'tools', 'standins'))
sys.path.insert(0, REF)
sys.path.insert(0, REPO)
os.chdir(REF)
from environment import MazeEnv  # noqa: E402
from gnnmp.maze2d import Maze3D  # noqa: E402

ref = MazeEnv(dim=3, map_file='maze_files/mazes_hard_3.npz')
with np.load('maze_files/mazes_hard_3.npz') as f:
    mine = Maze3D(f['maps'], f['init_states'], f['goal_states'])
rng = np.random.default_rng(0)
bad = 0
for prob in range(6):
    ref.init_new_problem(prob); mine.init_new_problem(prob)
    for _ in range(400):
        a = rng.uniform(-1, 1, 3).astype(np.float32) * np.array([1, 1, 0.4], np.float32)
        b = (a + rng.normal(0, 0.15, 3).astype(np.float32)).astype(np.float32)
        b[2] = np.float32(((b[2] + 0.4) % 0.8) - 0.4)
        d0, d1 = ref.distance(a.copy(), b.copy()), mine.distance(a.copy(), b.copy())
        i0, i1 = ref.interpolate(a.copy(), b.copy(), 0.37), mine.interpolate(a.copy(), b.copy(), 0.37)
        ref.collision_check_count = mine.collision_check_count = 0
        e0, e1 = ref._edge_fp(a.copy(), b.copy()), mine._edge_fp(a.copy(), b.copy())
        c0, c1 = ref.collision_check_count, mine.collision_check_count
        s0, s1 = ref._state_fp(a.copy()), mine._state_fp(a.copy())
        ok = np.array_equal(np.asarray(d0), np.asarray(d1)) and np.array_equal(i0, i1) and i0.dtype == i1.dtype and bool(e0) == bool(e1) \
            and c0 == c1 and bool(s0) == bool(s1)
        bad += not ok
        if not ok and bad < 5:
            print('MISMATCH', prob, a, b, d0, d1, i0, i1, e0, e1, c0, c1)
print('queries 2400, mismatches', bad)
sys.exit(1 if bad else 0)

import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import gnnmp
from gnnmp import planner
from gnnmp.maze2d import Maze2D
from gnnmp.weights import load_weights
with np.load(os.path.join(REPO, 'tests', 'golden', 'evalset_mazehard_first1000.npz')) as f:
    env = Maze2D(f['maps'], f['init_states'], f['goal_states'])
m = gnnmp.EncoderProcessDecoder(2, 2, 32, 2).eval(); m.load_state_dict(load_weights('weights_maze'))
ms = gnnmp.ModelSmoother(workspace_size=2, config_size=2, embed_size=128, obs_size=6).eval(); ms.load_state_dict(load_weights('smooth_2d_attv3'))
def run(n, chunk, workers=1):
    rows = []
    planner.eval_gnn_device(env, range(n), m, ms, device='cuda:0', chunk=chunk, workers=workers, rows_out=rows)
    return np.array(rows, dtype=np.float64)
A = run(300, 1024)
for chunk in (256, 128, 64, 32):
    B = run(300, chunk)
    bad = [i for i in range(300) if not np.array_equal(A[i, [0, 3, 5, 6]], B[i, [0, 3, 5, 6]])]
    print('chunk', chunk, 'explore mismatches', len(bad), bad[:12], 'smooth mismatches', int((A[:, 4] != B[:, 4]).sum()), flush=True)

import os, sys, time, cProfile, pstats
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import numpy as np, torch, gnnmp
from gnnmp import planner
from gnnmp.maze2d import Maze2D
from gnnmp.weights import load_weights
dev = torch.device('cuda:0')
with np.load(os.path.join(REPO, 'tests', 'golden', 'evalset_mazehard_first1000.npz')) as f:
    env = Maze2D(f['maps'], f['init_states'], f['goal_states'])
m = gnnmp.EncoderProcessDecoder(2, 2, 32, 2).eval()
m.load_state_dict(load_weights('weights_maze'))
ms = gnnmp.ModelSmoother(workspace_size=2, config_size=2, embed_size=128, obs_size=6).eval()
ms.load_state_dict(load_weights('smooth_2d_attv3'))
np.random.seed(1234)
for i in range(2):
    env.init_new_problem(i)
    planner.explore(env, m, ms, True, batch=500, t_max=500, k=30, device=dev)
pr = cProfile.Profile()
fw = 0
pr.enable()
for i in range(2, 10):
    env.init_new_problem(i)
    r = planner.explore(env, m, ms, True, batch=500, t_max=500, k=30, device=dev)
    fw += r['forward']
pr.disable()
print('forward ms per problem', fw / 8 * 1e3)
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)

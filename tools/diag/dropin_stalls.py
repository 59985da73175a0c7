"""Which part of the host loop provokes the sporadic ~60 ms stall of the drop-in forward span?  Variants over the same
problems; every forward span > 5 ms is printed with the thread's CPU time and context-switch / page-fault deltas across it.
python tools/diag/dropin_stalls.py [problems-per-variant]"""
import os, resource, sys, time
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
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
T0 = time.perf_counter()


def run(tag, **kw):
    np.random.seed(1234)
    spans = []
    for i in range(n):
        env.init_new_problem(i)
        r = planner.explore(env, m, ms, True, batch=500, t_max=500, k=30, device=dev, **kw)
        s = r['forward_split']
        spans.append(r['forward'] * 1e3)
        if r['forward'] > 5e-3:
            print('   %-22s problem %3d at %.2f s: forward %.1f ms (obs %.1f h2d %.1f call %.1f d2h %.1f)' % (
                tag, i, time.perf_counter() - T0, 1e3 * r['forward'], 1e3 * s['obs_data'], 1e3 * s['h2d'], 1e3 * s['module_call'],
                1e3 * s['d2h_wait']), flush=True)
    spans.sort()
    print('%-24s mean %.3f  median %.3f  p90 %.3f  max %.1f  >5 ms: %d of %d' % (
        tag, sum(spans) / n, spans[n // 2], spans[int(.9 * n)], spans[-1], sum(x > 5 for x in spans), n), flush=True)


for i in range(3):
    env.init_new_problem(i); planner.explore(env, m, ms, True, batch=500, t_max=500, k=30, device=dev)
run('baseline')
run('no smoother', smoother='none')
run('sparse', sparse=True)
run('sparse, no smoother', sparse=True, smoother='none')
run('baseline again')

"""Where the time of the reference's own call goes (eval_gnn.py:193-196: obs_data + H2D + model(**kw) + .cpu()), in a FRESH
process and again after the allocator states bench.py leaves behind.  python tools/diag/dropin_split.py"""
import os, sys, time
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


def run(tag, n=16, **kw):
    np.random.seed(1234)
    for i in range(3):
        env.init_new_problem(i); planner.explore(env, m, ms, True, batch=500, t_max=500, k=30, device=dev, **kw)
    fwd, split = 0., {}
    for i in range(n):
        env.init_new_problem(i)
        r = planner.explore(env, m, ms, True, batch=500, t_max=500, k=30, device=dev, **kw)
        fwd += r['forward']
        for k_, v_ in r['forward_split'].items():
            split[k_] = split.get(k_, 0) + v_
    print('%-34s forward %.3f ms/problem  %s' % (tag, 1e3 * fwd / n, {k_: round(1e3 * v_ / n, 3) for k_, v_ in split.items()}), flush=True)


run('fresh process')
run('again')
torch.cuda.empty_cache()
run('after empty_cache')
m.status_checks = False; ms.status_checks = False
run('status_checks off')
m.status_checks = True; ms.status_checks = True
run('sparse', sparse=True)
# a 1 GB block through the caching allocator and back out, like bench.py's dense leg
x = torch.empty(256 * 1000 * 1000, device=dev); del x
run('after 1 GB alloc (cached)')
torch.cuda.empty_cache()
run('after 1 GB alloc + empty_cache')
# device planner (threads, streams, 45 GB reserved) then the host loop again
idx = list(range(256))
planner.eval_gnn_device(env, idx, m, ms, device=dev)
run('after eval_gnn_device')
torch.cuda.empty_cache()
run('after eval_gnn_device + empty_cache')
# raw pieces at N = 1002
P = torch.empty(1002, 1002, device=dev)
for tag, fn in (('P.cpu() pageable 4 MB', lambda: P.cpu()),
                ('P -> pinned (non_blocking + sync)', None)):
    if fn is None:
        pin = torch.empty(1002, 1002, pin_memory=True)
        fn = lambda: (pin.copy_(P, non_blocking=True), torch.cuda.synchronize())
    for _ in range(5):
        fn()
    t = time.perf_counter()
    for _ in range(50):
        fn()
    print('%-34s %.3f ms' % (tag, (time.perf_counter() - t) / 50 * 1e3))

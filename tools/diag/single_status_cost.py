#!/usr/bin/env python
"""Run ON THE GPU BOX: what the per-forward status slot costs the reference's one-graph call (1000-node maze2 graph, back-to-back calls)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import torch, gnnmp
from gnnmp.synth import ENVS, synth_graph
from gnnmp.weights import load_weights
dev = torch.device('cuda:0')
e = ENVS['maze2']
m = gnnmp.EncoderProcessDecoder(e['workspace'], e['C'], e['d'], e['S']).eval()
m.load_state_dict(load_weights(e['ckpt']))
g = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synth_graph('maze2', 1000, 8, seed=1).items()}
b1 = m._single(g['goal'], g['v'], g['obstacles'], g['edge_index'])


def med(fn, n=50):
    for _ in range(10):
        fn()
    ts = []
    for _ in range(7):
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t) / n * 1e6)
    return sorted(ts)[3]


for checks in (True, False, True, False):
    m.status_checks = checks
    print('status_checks=%-5s sparse %.1f us   dense drop-in %.1f us' % (
        checks, med(lambda: m.forward_batch(b1, 5)),
        med(lambda: m(goal=g['goal'], loop=5, v=g['v'], obstacles=g['obstacles'], edge_index=g['edge_index']))), flush=True)

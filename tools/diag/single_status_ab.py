import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'tests'))
import torch, gnnmp
from gnnmp.weights import load_weights
from gnnmp.synth import ENVS, synth_graph
dev = torch.device('cuda:0')
e = ENVS['maze2']
g = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synth_graph('maze2', 1000, 8, seed=1).items()}
m = gnnmp.EncoderProcessDecoder(e['workspace'], e['C'], e['d'], e['S']).eval(); m.load_state_dict(load_weights(e['ckpt']))
def med(fn, n=50):
    for _ in range(10): fn()
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t) / n)
    return sorted(ts)[2] * 1e6
for rep in range(2):
    for sc in (True, 'always', False):
        m.status_checks = sc
        a = med(lambda: m.edge_scores(g['goal'], 5, g['v'], g['obstacles'], g['edge_index']))
        b = med(lambda: m(goal=g['goal'], loop=5, v=g['v'], obstacles=g['obstacles'], edge_index=g['edge_index']))
        print('status_checks %-7s: sparse %.1f us  dense drop-in %.1f us' % (sc, a, b), flush=True)

import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import torch, gnnmp
from gnnmp.weights import load_weights
from gnnmp.synth import ENVS, synth_graph
dev = torch.device('cuda:0')
def timeit(fn, n=50, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for env, n, k in (('maze2', 200, 6), ('maze2', 1000, 8), ('maze2', 1002, 41), ('kuka7', 2000, 10)):
    e = ENVS[env]
    m = gnnmp.EncoderProcessDecoder(e['workspace'], e['C'], e['d'], e['S']); m.load_state_dict(load_weights(e['ckpt']))
    g = {kk: (v.to(dev) if torch.is_tensor(v) else v) for kk, v in synth_graph(env, n, k).items()}
    b = m._single(g['goal'], g['v'], g['obstacles'], g['edge_index'])
    fb = timeit(lambda: m.forward_batch(b, 5))
    m.profile(dev, True)
    for _ in range(20): m.forward_batch(b, 5)
    pr = m.profile_read(dev); m.profile(dev, False)
    graph, _ = m.capture(b, 5)
    rep = timeit(graph.replay)
    print('%-6s N=%-5d E=%-6d prebuilt batch %.3f ms | hipGraph replay %.3f ms | stages us: %s' % (env, n, g['edge_index'].shape[1], fb * 1e3, rep * 1e3, {k_: round(v[0] / 20 * 1e3, 1) for k_, v in pr.items()}))

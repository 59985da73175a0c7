"""single smoother call (the reference's pattern: loop = 1, one problem): drop-in call vs prebuilt batch"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import torch, gnnmp
from gnnmp.weights import load_weights
from gnnmp.planner import chain_edge_index
from gnnmp.smoother import SmoothBatch
dev = torch.device('cuda:0')
ms = gnnmp.ModelSmoother(workspace_size=2, config_size=2, embed_size=128, obs_size=6).eval()
ms.load_state_dict(load_weights('smooth_2d_attv3'))
gen = torch.Generator().manual_seed(3)
P, F, Co = 20, 500, 500
path = (torch.rand(P, 2, generator=gen) * 2 - 1).to(dev)
free = (torch.rand(F, 2, generator=gen) * 2 - 1).to(dev)
coll = (torch.rand(Co, 2, generator=gen) * 2 - 1).to(dev)
ei = chain_edge_index(P).to(dev)
one = SmoothBatch([path], [free], [coll], [ei], dev)


def timeit(fn, n=100):
    for _ in range(10):
        fn()
    ts = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / n)
    return sorted(ts)[2] * 1e6


print('prebuilt batch: %.1f us   drop-in call: %.1f us' % (
    timeit(lambda: ms.forward_batch(one, 1)), timeit(lambda: ms(path=path, free=free, collided=coll, obstacles=None, edge_index=ei, loop=1))))

#!/usr/bin/env python
"""Run ON THE GPU BOX: smoother fp32 batched forward (C = 14, 20 waypoints, 500 + 500 samples, loop 1) at several batch sizes; run once with
GNNMP_SM_STREAM=0 and once with =1 (the switch is read once per process)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import torch, gnnmp
from gnnmp.weights import load_weights
from gnnmp.planner import chain_edge_index
from gnnmp.smoother import SmoothBatch
dev = torch.device('cuda:0')
C = 14
ms = gnnmp.ModelSmoother(workspace_size=3, config_size=C, embed_size=128, obs_size=6).eval()
ms.load_state_dict(load_weights('smooth_14d_attv3'))
gen = torch.Generator().manual_seed(3)
mk = lambda n: (torch.rand(n, C, generator=gen) * 2 - 1)          # noqa: E731
for B in [int(a) for a in sys.argv[1:]] or [128, 256, 512, 1024, 2048, 4096]:
    sb = SmoothBatch([mk(20) for _ in range(B)], [mk(500) for _ in range(B)], [mk(500) for _ in range(B)], [chain_edge_index(20)] * B, dev)
    for _ in range(5):
        ms.forward_batch(sb, 1)
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            ms.forward_batch(sb, 1)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 10 * 1e3)
    print('GNNMP_SM_STREAM=%s  B=%5d  %.3f ms  %.0f calls/s' % (os.environ.get('GNNMP_SM_STREAM', 'auto'), B, sorted(ts)[2], B / sorted(ts)[2] * 1e3), flush=True)
    del sb

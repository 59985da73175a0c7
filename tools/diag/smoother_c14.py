#!/usr/bin/env python
"""Diagnostic: smoother batch-of-256 timing per (C, dtype) with per-launch spread."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'tests'))
import torch, gnnmp
from gnnmp.weights import load_weights
from gnnmp.planner import chain_edge_index
from gnnmp.smoother import SmoothBatch
dev = torch.device('cuda:0')
only = sys.argv[1:]
for name, C in (('smooth_2d_attv3', 2), ('smooth_7d_attv3', 7), ('smooth_13d_attv3', 13), ('smooth_14d_attv3', 14)):
    for dtype in ('fp32', 'bf16'):
        if only and ('%d%s' % (C, dtype)) not in only:
            continue
        gen = torch.Generator().manual_seed(3)
        ms = gnnmp.ModelSmoother(workspace_size=3, config_size=C, embed_size=128, obs_size=6, scale=1.0).eval()
        ms.load_state_dict(load_weights(name)); ms.mlp_dtype = dtype
        mk = lambda n: (torch.rand(n, C, generator=gen) * 2 - 1)
        B = 256
        many = SmoothBatch([mk(20) for _ in range(B)], [mk(500) for _ in range(B)], [mk(500) for _ in range(B)],
                           [chain_edge_index(20)] * B, dev)
        ts = []
        for i in range(12):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ms.forward_batch(many, 1)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        print(name, C, dtype, ' '.join('%.3f' % t for t in ts), flush=True)

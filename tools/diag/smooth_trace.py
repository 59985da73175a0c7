"""Smoother batch in a loop for a rocprofv3 --kernel-trace --stats pass: python tools/diag/smooth_trace.py [name C B dtype]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
import gnnmp
from gnnmp.weights import load_weights
from gnnmp.planner import chain_edge_index
from gnnmp.smoother import SmoothBatch

name = sys.argv[1] if len(sys.argv) > 1 else 'smooth_14d_attv3'
C = int(sys.argv[2]) if len(sys.argv) > 2 else 14
B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
dtype = sys.argv[4] if len(sys.argv) > 4 else 'fp32'
dev = torch.device('cuda', 0)
gen = torch.Generator().manual_seed(3)
ms = gnnmp.ModelSmoother(workspace_size=3, config_size=C, embed_size=128, obs_size=6).eval()
ms.load_state_dict(load_weights(name))
ms.mlp_dtype = dtype
mk = lambda n: (torch.rand(n, C, generator=gen) * 2 - 1)          # noqa: E731
many = SmoothBatch([mk(20) for _ in range(B)], [mk(500) for _ in range(B)], [mk(500) for _ in range(B)],
                   [chain_edge_index(20)] * B, dev)
for _ in range(30):
    ms.forward_batch(many, 1)
torch.cuda.synchronize()

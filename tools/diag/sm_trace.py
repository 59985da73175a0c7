#!/usr/bin/env python
"""Run ON THE GPU BOX with a -DGNNMP_SM_TRACE build: per-workgroup timeline of the smoother's split message launch (batch of problems,
fp32): start / source half done / flags up / target rows in hand / first layer done / end, by role (target tiles vs edge tiles).
    GNNMP_LIB=.../libgnnmp_smtrace.so python tools/diag/sm_trace.py [name C B dtype]"""
import ctypes, os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import gnnmp
from gnnmp import _lib
from gnnmp.weights import load_weights
from gnnmp.planner import chain_edge_index
from gnnmp.smoother import SmoothBatch
name, C, B, dtype = (sys.argv[1:5] + ['smooth_14d_attv3', '14', '256', 'fp32'][len(sys.argv) - 1:])[:4]
C, B = int(C), int(B)
dev = torch.device('cuda', 0)
gen = torch.Generator().manual_seed(3)
ms = gnnmp.ModelSmoother(workspace_size=3, config_size=C, embed_size=128, obs_size=6).eval()
ms.load_state_dict(load_weights(name)); ms.mlp_dtype = dtype
mk = lambda n: (torch.rand(n, C, generator=gen) * 2 - 1)          # noqa: E731
many = SmoothBatch([mk(20) for _ in range(B)], [mk(500) for _ in range(B)], [mk(500) for _ in range(B)], [chain_edge_index(20)] * B, dev)
for _ in range(5):
    ms.forward_batch(many, 1)
torch.cuda.synchronize()
L = _lib.lib()
n = 8192
buf = np.zeros(8 * n, dtype=np.int64)
assert L.gnnmp_debug_sm_trace(buf.ctypes.data_as(ctypes.c_void_p), n) == 0
t = buf.reshape(-1, 8)
t = t[t[:, 0] > 0]
t0 = t[:, 0].min()
us = lambda x: (x - t0) / 100.0
role = t[:, 7]
print('%s C=%d B=%d %s: %d workgroups recorded (%d target tiles, %d edge tiles); launch span %.1f us' % (
    name, C, B, dtype, len(t), int((role == 1).sum()), int((role == 0).sum()), us(t[:, 5].max())))
tt, te = t[role == 1], t[role == 0]
pc = lambda x: tuple(np.percentile(x, q) for q in (10, 50, 90, 100))
print('target tiles: start p10 %.1f p50 %.1f p90 %.1f max %.1f | length p10 %.1f p50 %.1f p90 %.1f max %.1f us' % (pc(us(tt[:, 0])) + pc((tt[:, 5] - tt[:, 0]) / 100.0)))
print('edge tiles:   start p10 %.1f p50 %.1f p90 %.1f max %.1f | length p10 %.1f p50 %.1f p90 %.1f max %.1f us' % (pc(us(te[:, 0])) + pc((te[:, 5] - te[:, 0]) / 100.0)))
for a, b, what in ((0, 1, 'source half (node_code of the sources + exchange)'), (1, 2, 'waiting for the target tiles\' flags'), (2, 3, 'target rows in hand'),
                   (3, 4, 'first layer (W_src x_j) + relu'), (4, 5, 'second layer + store')):
    d = (te[:, b] - te[:, a]) / 100.0
    print('   edge tile, %-52s p10 %.1f p50 %.1f p90 %.1f max %.1f us' % ((what,) + pc(d)))
wait = (te[:, 2] - te[:, 1]) / 100.0
order = np.argsort(te[:, 0])
print('   flag wait of the first 1024 edge tiles to start: p50 %.1f us; of the rest: p50 %.1f us' % (np.percentile(wait[order[:1024]], 50), np.percentile(wait[order[1024:]], 50)))
start, end = us(t[:, 0]), us(t[:, 5])
grid = np.linspace(0, end.max(), 21)
print('workgroups alive at t (us): ' + ' '.join('%.0f:%d' % (x, ((start <= x) & (end > x)).sum()) for x in grid))
hw, xcc = t[:, 6] & 0xffffffff, t[:, 6] >> 32
cu = ((hw >> 8) & 15) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (xcc << 8)
print('distinct (XCC, SE, SH, CU) ids seen: %d; workgroups per id p50 %d max %d' % (len(np.unique(cu)), np.percentile(np.bincount(np.unique(cu, return_inverse=True)[1]), 50), np.bincount(np.unique(cu, return_inverse=True)[1]).max()))

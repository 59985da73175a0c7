#!/usr/bin/env python
"""Run ON THE GPU BOX with a -DGNNMP_SM_TRACE build: where wave 0 of every workgroup of sm_msg_stream_kernel spends its cycles per
column -- the vmcnt wait, the barrier, DMA issue + operand reads + MFMA issue.   GNNMP_LIB=.../libgnnmp_smtrace.so python tools/diag/sm_stream_trace.py"""
import ctypes, os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import gnnmp
from gnnmp import _lib
from gnnmp.weights import load_weights
from gnnmp.planner import chain_edge_index
from gnnmp.smoother import SmoothBatch
C, B = 14, int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device('cuda', 0)
gen = torch.Generator().manual_seed(3)
ms = gnnmp.ModelSmoother(workspace_size=3, config_size=C, embed_size=128, obs_size=6).eval()
ms.load_state_dict(load_weights('smooth_14d_attv3'))
mk = lambda n: (torch.rand(n, C, generator=gen) * 2 - 1)          # noqa: E731
many = SmoothBatch([mk(20) for _ in range(B)], [mk(500) for _ in range(B)], [mk(500) for _ in range(B)], [chain_edge_index(20)] * B, dev)
for _ in range(5):
    ms.forward_batch(many, 1)
torch.cuda.synchronize()
n = 8192
buf = np.zeros(8 * n, dtype=np.int64)
assert _lib.lib().gnnmp_debug_sm_trace(buf.ctypes.data_as(ctypes.c_void_p), n) == 0
t16 = buf.reshape(-1, 16)
t16 = t16[t16[:, 7] == 2]
t = t16[:, :8]
m = t16[:, 8:15]
life = t[:, 5] - t[:, 0]
print('%d workgroups; life (cycles) p10 %.0f p50 %.0f p90 %.0f max %.0f; span first start -> last end %.0f cycles' % (
    (len(t),) + tuple(np.percentile(life, q) for q in (10, 50, 90, 100)) + (t[:, 5].max() - t[:, 0].min(),)))
for grp, sel in (('more than one edge round', t[:, 4] > 14), ('one edge round', t[:, 4] <= 14)):
    g = t[sel]
    if not len(g):
        continue
    print('%-28s n %4d  columns %.1f  per column: vmcnt wait %.0f  barrier %.0f  issue+reads+mfma %.0f  | life %.0f  outside the columns %.0f' % (
        grp, len(g), g[:, 4].mean(), (g[:, 1] / g[:, 4]).mean(), (g[:, 2] / g[:, 4]).mean(), (g[:, 3] / g[:, 4]).mean(),
        (g[:, 5] - g[:, 0]).mean(), ((g[:, 5] - g[:, 0]) - g[:, 1] - g[:, 2] - g[:, 3]).mean()))
for grp, sel in (('more rounds', t[:, 4] > 14), ('one edge round', t[:, 4] <= 14)):
    g, mm = t[sel], m[sel]
    if not len(g):
        continue
    d = lambda a, b: (mm[:, b] - mm[:, a]).mean()          # noqa: E731
    print('%-16s LAST edge round, cycles: kernel start -> round start %.0f | setup loads %.0f | source half %.0f | flag wait + consensus %.0f | target rows %.0f | W_src %.0f | W_02 + store %.0f | after -> end %.0f' % (
        grp, (mm[:, 0] - g[:, 0]).mean(), d(0, 1), d(1, 2), d(2, 3), d(3, 4), d(4, 5), d(5, 6), (g[:, 5] - mm[:, 6]).mean()))

#!/usr/bin/env python
"""Run ON THE GPU BOX with a -DGNNMP_MP_TRACE build (GNNMP_LIB=.../libgnnmp_trace.so): per-wave timeline of the LAST mp_fused
launch of one forward -- when every wave starts, how long its tiles' edge / node phases take, when it ends -- i.e. where the
wave slots of the launch sit idle.   python tools/diag/mp_trace.py [env nodes k1 graphs dtype]"""
import ctypes, os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import gnnmp
from gnnmp import _lib
from gnnmp.synth import ENVS, synth_batch_gpu
from gnnmp.weights import load_weights
env, nodes, k1, G, dt = (sys.argv[1:6] + ['kuka7', '2000', '10', '64', 'bf16'][len(sys.argv) - 1:])[:5]
e = ENVS[env]
graphs = synth_batch_gpu(env, int(nodes), int(k1), int(G), 'cuda:0')
m = gnnmp.EncoderProcessDecoder(e['workspace'], e['C'], e['d'], e['S']).eval()
m.load_state_dict(load_weights(e['ckpt'])); m.mlp_dtype = dt
b = gnnmp.GraphBatch.from_graphs(graphs, e['S'], 'cuda:0')
for _ in range(3):
    m.forward_batch(b, 5)
torch.cuda.synchronize()
L = _lib.lib()
L.gnnmp_debug_mp_trace.restype = ctypes.c_longlong
buf = np.zeros(1 << 22, dtype=np.int64)
n = L.gnnmp_debug_mp_trace(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_longlong(buf.size))
t = buf[:n].reshape(-1, 8).astype(np.float64)
t = t[t[:, 0] > 0]
t0 = t[:, 0].min()
us = lambda x: (x - t0) / 100.0           # 100 MHz
start = us(t[:, 0])
last = np.array([row[row > 0].max() for row in t])
end = us(last)
ntile = ((t > 0).sum(1) - 1) // 5
print('%s N=%s k1=%s x%s %s: %d waves recorded, launch span %.1f us' % (env, nodes, k1, G, dt, len(t), end.max()))
print('wave start  (us): min %.1f  p50 %.1f  p90 %.1f  max %.1f' % (start.min(), np.percentile(start, 50), np.percentile(start, 90), start.max()))
print('wave end    (us): min %.1f  p10 %.1f  p50 %.1f  p90 %.1f  max %.1f' % (end.min(), np.percentile(end, 10), np.percentile(end, 50), np.percentile(end, 90), end.max()))
dur = end - start
print('wave length (us): min %.1f  p50 %.1f  p90 %.1f  max %.1f ; tiles per wave: %s' % (dur.min(), np.percentile(dur, 50), np.percentile(dur, 90), dur.max(), np.bincount(ntile.astype(int)).tolist()))
full = t[:, 5] > 0
if full.any():                       # builds with the two extra marks inside the node phase: [3] H done, [4] Y done, [5] end
    tt = t[full]
    print('node phase of the first tile (us): X rows + Wlx + Wla -> H  p50 %.1f | R rows + M1 -> Y  p50 %.1f | stores + M2 (+ M3)  p50 %.1f' % (
        np.percentile((tt[:, 3] - tt[:, 2]) / 100.0, 50), np.percentile((tt[:, 4] - tt[:, 3]) / 100.0, 50), np.percentile((tt[:, 5] - tt[:, 4]) / 100.0, 50)))
    t = np.concatenate((t[:, :3], t[:, 5:6], t[:, 6:], np.zeros((len(t), 2))), axis=1)[:, :8]     # back to [start, tile start, edge end, node end, ...]
has = t[:, 3] > 0
if has.any():
    tt = t[has]
    pro = us(tt[:, 1]) - us(tt[:, 0]); edge = (tt[:, 2] - tt[:, 1]) / 100.0; node = (tt[:, 3] - tt[:, 2]) / 100.0
    print('first tile of a wave (us): prologue p50 %.1f p90 %.1f | edge phase p50 %.1f p90 %.1f max %.1f | node phase p50 %.1f p90 %.1f max %.1f' % (
        np.percentile(pro, 50), np.percentile(pro, 90), np.percentile(edge, 50), np.percentile(edge, 90), edge.max(), np.percentile(node, 50), np.percentile(node, 90), node.max()))
busy = dur.sum()
print('wave-time / (span x waves resident at once [1024 SIMDs x 2]) = %.3f' % (busy / (end.max() * 2048)))
# occupancy over time
grid = np.linspace(0, end.max(), 23)
occ = [((start <= x) & (end > x)).sum() for x in grid]
print('waves alive at t (us): ' + ' '.join('%.0f:%d' % (x, o) for x, o in zip(grid, occ)))

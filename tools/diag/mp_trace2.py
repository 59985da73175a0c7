#!/usr/bin/env python
"""Run ON THE GPU BOX with a -DGNNMP_MP_TRACE build: per-wave, per-tile timeline of the LAST mp_fused launch of one forward (32 slots per
wave: start, up to five tiles x (start, edge end, H, Y, node end), HW_ID / XCC_ID, the first two tile ids, wave end) joined with the
tiles' edge counts: where the launch's time goes and which waves finish late.   python tools/diag/mp_trace2.py [env nodes k1 graphs dtype]"""
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
buf = np.zeros(1 << 24, dtype=np.int64)
n = L.gnnmp_debug_mp_trace(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_longlong(buf.size))
t = buf[:n].reshape(-1, 32)
t = t[t[:, 0] > 0]
W = len(t)
t0 = t[:, 0].min()
us = lambda x: (x - t0) / 100.0
# per-tile edge counts in the padded node space (kPad = 256 rows per graph block)
deg_tiles = []
for g in graphs:
    N = g['v'].shape[0]
    d = torch.bincount(g['edge_index'][1].cpu(), minlength=N).numpy()
    Np = (N + 255) // 256 * 256
    dp = np.zeros(Np, dtype=np.int64); dp[:N] = d
    deg_tiles.append(dp.reshape(-1, 32).sum(1))
deg_tiles = np.concatenate(deg_tiles)
start, end = us(t[:, 0]), us(t[:, 31])
hw, xcc = t[:, 28] & 0xffffffff, t[:, 28] >> 32
simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
print('%s N=%s k1=%s x%s %s: %d waves, launch span (first start .. last wave end) %.1f us' % (env, nodes, k1, G, dt, W, end.max()))
print('wave end (us): min %.1f p10 %.1f p25 %.1f p50 %.1f p75 %.1f p90 %.1f max %.1f' % tuple(np.percentile(end, q) for q in (0, 10, 25, 50, 75, 90, 100)))
ntile = np.array([int((row[1:26:5] > 0).sum()) for row in t])
print('tiles per wave:', np.bincount(ntile).tolist())
# per tile records
rec = []
for w in range(W):
    for k in range(5):
        s = t[w, 1 + 5 * k: 6 + 5 * k]
        if s[0] <= 0 or s[4] <= 0:
            continue
        tid = int(t[w, 29 + k]) if k < 2 else -1
        rec.append((w, k, tid, (s[1] - s[0]) / 100.0, (s[2] - s[1]) / 100.0, (s[3] - s[2]) / 100.0, (s[4] - s[3]) / 100.0,
                    deg_tiles[tid] if 0 <= tid < len(deg_tiles) else -1))
rec = np.array(rec, dtype=np.float64)
for k in range(int(rec[:, 1].max()) + 1):
    r = rec[rec[:, 1] == k]
    edges = r[:, 7]
    has = edges >= 0
    line = 'tile #%d of a wave: n %d | edge phase p50 %.1f p90 %.1f | H p50 %.1f | Y p50 %.1f | rest p50 %.1f' % (
        k, len(r), np.percentile(r[:, 3], 50), np.percentile(r[:, 3], 90), np.percentile(r[:, 4], 50), np.percentile(r[:, 5], 50), np.percentile(r[:, 6], 50))
    if has.any():
        ch = np.ceil(edges[has] / 32.0)
        line += ' | edges p50 %d p90 %d | us per 32-edge chunk p50 %.2f p90 %.2f' % (np.percentile(edges[has], 50), np.percentile(edges[has], 90),
                                                                                     np.percentile(r[has, 3] / np.maximum(ch, 1), 50), np.percentile(r[has, 3] / np.maximum(ch, 1), 90))
    print(line)
# gaps: wave time not inside any tile (prologue, between tiles, after the last tile)
tile_time = np.zeros(W)
for row in rec:
    tile_time[int(row[0])] += row[3:7].sum()
print('wave length p50 %.1f | inside tiles p50 %.1f | outside (staging, tile starts, drain) p50 %.1f' % (
    np.percentile(end - start, 50), np.percentile(tile_time, 50), np.percentile(end - start - tile_time, 50)))
# who finishes late?
late = end > np.percentile(end, 60)
def by(name, key):
    ks = np.unique(key)
    print('late-wave share by %s: ' % name + ' '.join('%d:%.2f' % (k, late[key == k].mean()) for k in ks))
by('XCC', xcc); by('SE', se); by('SIMD', simd); by('wave-in-WG', np.arange(W) % 4)
wg = np.arange(W) // 4
wg_end = np.array([end[wg == i].max() for i in range(wg.max() + 1)])
print('workgroup end (us): p10 %.1f p50 %.1f p90 %.1f max %.1f ; spread inside a workgroup p50 %.1f' % (
    np.percentile(wg_end, 10), np.percentile(wg_end, 50), np.percentile(wg_end, 90), wg_end.max(),
    np.percentile([end[wg == i].max() - end[wg == i].min() for i in range(wg.max() + 1)], 50)))
# edges of a wave's tiles vs its end
e0 = np.array([deg_tiles[int(x)] if 0 <= x < len(deg_tiles) else 0 for x in t[:, 29]])
e1 = np.array([deg_tiles[int(x)] if 0 < x < len(deg_tiles) else 0 for x in t[:, 30]])
tot = e0 + e1
print('edges of the first two tiles per wave: p10 %d p50 %d p90 %d ; correlation with wave end %.2f ; late waves mean %d, early waves mean %d' % (
    np.percentile(tot, 10), np.percentile(tot, 50), np.percentile(tot, 90), np.corrcoef(tot, end)[0, 1], tot[late].mean(), tot[~late].mean()))
busy = (end - start).sum()
print('wave-time / (span x resident waves) = %.3f' % (busy / (end.max() * W)))
grid = np.linspace(0, end.max(), 23)
print('waves alive at t (us): ' + ' '.join('%.0f:%d' % (x, ((start <= x) & (end > x)).sum()) for x in grid))

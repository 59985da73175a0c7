#!/usr/bin/env python
"""Run ON THE GPU BOX: the sweep gnnmp.dist's cost model is fitted from -- device time of ONE batched explorer forward per family at
1 ... 128 problems (1000-node k1 = 8 RGGs, loop 5), resident inputs, median of 3 x 10 forwards -- and the
least-squares fit  time_ms = a(d, dtype) + b(d, dtype) x GFLOP  (GFLOP = reference-formulation FLOPs of the batch).
Kept as profiles/rNN_cost_sweep.txt; the fitted constants go into gnn-motion-planning_amd/dist.py (_COST_FIT)."""
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import gnnmp  # noqa: E402
from gnnmp.dist import reference_flops  # noqa: E402
from gnnmp.synth import ENVS, synth_batch_gpu  # noqa: E402
from gnnmp.weights import load_weights  # noqa: E402

DEV = 'cuda:0'
FAMILIES = ['maze2', 'snake7', 'ur5', 'kuka7', 'kuka14']
SIZES = (1, 2, 4, 8, 16, 32, 64, 128)
N, K1, LOOP = 1000, 8, 5
rows = []
for dtype in ('fp32', 'bf16'):
    for fi, env in enumerate(FAMILIES):
        e = ENVS[env]
        m = gnnmp.EncoderProcessDecoder(e['workspace'], e['C'], e['d'], e['S']).eval()
        m.load_state_dict(load_weights(e['ckpt']), strict=True)
        m.mlp_dtype = dtype
        graphs = synth_batch_gpu(env, N, K1, max(SIZES), DEV, seed0=5000 + 1000 * fi)
        for n in SIZES:
            b = gnnmp.GraphBatch.from_graphs(graphs[:n], e['S'], DEV)
            for _ in range(5):
                m.forward_batch(b, LOOP)
            ts = []
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(10):
                    m.forward_batch(b, LOOP)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) / 10 * 1e3)
            gf = sum(reference_flops(g['v'].shape[0], g['edge_index'].shape[1], g['obstacles'].reshape(-1, e['S']).shape[0],
                                     e['C'], e['d'], e['S'], LOOP) for g in graphs[:n]) / 1e9
            rows.append((dtype, env, e['d'], n, sorted(ts)[1], gf))
            print('%-5s %-7s d=%d  n=%3d  %.4f ms  %.2f GFLOP' % rows[-1], flush=True)
fit = {}
for dtype in ('fp32', 'bf16'):
    for d in (32, 64):
        pts = [(r[5], r[4]) for r in rows if r[0] == dtype and r[2] == d]
        X = torch.tensor([[1.0, g] for g, _ in pts], dtype=torch.float64)
        y = torch.tensor([t for _, t in pts], dtype=torch.float64)
        sol = torch.linalg.lstsq(X, y.unsqueeze(1)).solution.squeeze(1)
        pred = X @ sol
        fit['%d/%s' % (d, dtype)] = {'a_ms': round(float(sol[0]), 4), 'b_ms_per_gflop': round(float(sol[1]), 6),
                                     'max_rel_residual': round(float(((pred - y).abs() / y).max()), 3), 'points': len(pts)}
print('fit: time_ms = a + b x GFLOP per (d, dtype), families pooled')
print(json.dumps(fit, indent=1))
# the curves of gnnmp.dist._COST_CURVE: per (d, dtype), families pooled per batch size -> (mean GFLOP, mean ms)
curves = {}
for dtype in ('fp32', 'bf16'):
    for d in (32, 64):
        pts = []
        for n in SIZES:
            sel = [r for r in rows if r[0] == dtype and r[2] == d and r[3] == n]
            pts.append((round(sum(r[5] for r in sel) / len(sel), 1), round(sum(r[4] for r in sel) / len(sel), 3)))
        curves['%d/%s' % (d, dtype)] = pts
print('curves (GFLOP, ms):')
print(json.dumps(curves))
# per family as well (does one (d, dtype) pair of constants serve families with 5 and with 116 obstacles?)
for dtype in ('fp32', 'bf16'):
    for env in FAMILIES:
        pts = [(r[5], r[4]) for r in rows if r[0] == dtype and r[1] == env]
        X = torch.tensor([[1.0, g] for g, _ in pts], dtype=torch.float64)
        y = torch.tensor([t for _, t in pts], dtype=torch.float64)
        sol = torch.linalg.lstsq(X, y.unsqueeze(1)).solution.squeeze(1)
        print('  %-5s %-7s a %.4f ms  b %.6f ms/GFLOP' % (dtype, env, float(sol[0]), float(sol[1])))

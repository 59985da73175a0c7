#!/usr/bin/env python
"""Run ON THE GPU BOX: per-fixture parity table of the HIP forwards against the goldens recorded from the unmodified
reference (fp32 and fp64 runs) -> stdout (kept as profiles/rNN_parity.txt).  Columns: max|gpu-ref32|, max|gpu-ref64|,
the reference's own fp32-vs-fp64 error, the per-fixture bar max(1e-5, 1.25 own) (tests/parity_bar.py: absolute, bar64 =
it, bar32 = it + own), pass flags, whether the bare north_star figure (|gpu-ref32| <= 1e-5 everywhere) holds, how many
elements exceed 1e-5 against ref32 / against ref64, and how many fail a strict allclose(rtol 1e-5, atol 1e-5) against ref32.
GNNMP_NODE_F64=0 in the environment gives the all-fp32 kernels of round 2 (attribution column)."""
import glob
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))
import gnnmp  # noqa: E402
import parity_bar  # noqa: E402
from gnnmp.synth import ENVS  # noqa: E402
from gnnmp.weights import load_weights  # noqa: E402

DEV = 'cuda:0'
SM = {'smooth_2d_attv3': (2, 1.0), 'smooth_7d_attv3': (7, 1.0), 'smooth_ur5_attv3': (6, 2 * np.pi),
      'smooth_snake_attv3': (7, 1.0), 'smooth_13d_attv3': (13, 1.0), 'smooth_14d_attv3': (14, 1.0)}
modes = sys.argv[1:] or ['fp32']
FMT = '%-44s %-7s %7d %10.3e %10.3e %10.3e %10.3e %5s %5s %9s %8d %8d %9d'
print('%-44s %-7s %7s %10s %10s %10s %10s %5s %5s %9s %8s %8s %9s' % ('fixture', 'mode', 'n', 'gpu-ref32', 'gpu-ref64', 'ref32-64',
                                                                      'bar atol', 'ok32', 'ok64', 'bare1e-5', '#>1e-5', '#>1e-5/64', '#allclose'))
for path in sorted(glob.glob(os.path.join(REPO, 'tests', 'golden', 'explorer_*.npz'))):
    with np.load(path) as f:
        r = {k: f[k] for k in f.files}
    env = os.path.basename(path).split('_')[1]
    e = ENVS[env]
    for mode in modes:
        m = gnnmp.EncoderProcessDecoder(e['workspace'], e['C'], e['d'], e['S'], use_obstacles=bool(r['use_obstacles'])).eval()
        m.load_state_dict(load_weights(e['ckpt']), strict=True)
        m.mlp_dtype = mode
        s = m.edge_scores(goal=torch.from_numpy(r['goal']).to(DEV), loop=int(r['loop']), v=torch.from_numpy(r['v']).to(DEV),
                          obstacles=torch.from_numpy(r['obstacles']).to(DEV),
                          edge_index=torch.from_numpy(r['edge_index']).to(DEV)).cpu()
        c = parity_bar.check(s, torch.from_numpy(r['scores_fp32']), torch.from_numpy(r['scores_fp64']))
        print(FMT % (os.path.basename(path)[:-4], mode, c['n'], c['err32'], c['err64'], c['own'], c['atol'], c['ok32'], c['ok64'],
                     c['bare_1e5'], c['n_over_1e5'], c['n_over_1e5_64'], c['n_fail_allclose']))
for path in sorted(glob.glob(os.path.join(REPO, 'tests', 'golden', 'smoother_*.npz'))):
    with np.load(path) as f:
        r = {k: f[k] for k in f.files}
    if 'out_fp64' not in r:            # the float32-kNN fixture has no fp64 twin (tests/test_smoother_parity.py covers it)
        continue
    name = os.path.basename(path).split('_P')[0].replace('smoother_', '')
    C, scale = SM[name]
    for mode in [m for m in modes if m != 'bf16x3']:
        m = gnnmp.ModelSmoother(workspace_size=3, config_size=C, embed_size=128, obs_size=6, scale=scale).eval()
        m.load_state_dict(load_weights(name), strict=True)
        m.mlp_dtype = mode
        out = m(path=torch.from_numpy(r['path']).to(DEV), free=torch.from_numpy(r['free']).to(DEV),
                collided=torch.from_numpy(r['collided']).to(DEV), obstacles=None,
                edge_index=torch.from_numpy(r['edge_index']).to(DEV), loop=int(r['loop'])).cpu()
        c = parity_bar.check(out.reshape(-1), torch.from_numpy(r['out_fp32']).reshape(-1), torch.from_numpy(r['out_fp64']).reshape(-1))
        print(FMT % (os.path.basename(path)[:-4], mode, c['n'], c['err32'], c['err64'], c['own'], c['atol'], c['ok32'], c['ok64'],
                     c['bare_1e5'], c['n_over_1e5'], c['n_over_1e5_64'], c['n_fail_allclose']))

"""bf16 operand mode of the smoother (BASELINE configs[4] lists "smoother GNN, bf16").  Same two bars as the
explorer's bf16 tests: (i) the kernels match a CPU emulation of their own formulation with bf16-rounded MFMA
operands; (ii) accuracy against the fp32 reference goldens: waypoint proposals are O(1) coordinates that the
planner only follows in RRT_EPS-limited, collision-checked steps (smoother.py:194-216), bar: max|d| <= 0.02 in
units of `scale` (measured <= 8e-3)."""
import os

import numpy as np
import pytest
import torch

from conftest import golden_files, load_weights
import gnnmp
from oracle import ref_bf16

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
CONF = {'smooth_2d_attv3': (2, 1.0), 'smooth_7d_attv3': (7, 1.0), 'smooth_ur5_attv3': (6, 2 * np.pi),
        'smooth_snake_attv3': (7, 1.0), 'smooth_13d_attv3': (13, 1.0), 'smooth_14d_attv3': (14, 1.0)}


# (the float32-kNN fixture is excluded: the emulation restates the float64-distance kNN of the default stand-in)
@pytest.mark.parametrize('path', [p for p in golden_files('smoother_') if 'knn32' not in p], ids=os.path.basename)
def test_bf16_smoother(path):
    with np.load(path) as f:
        r = {k: f[k] for k in f.files}
    name = os.path.basename(path).split('_P')[0].replace('smoother_', '')
    C, scale = CONF[name]
    m = gnnmp.ModelSmoother(workspace_size=3, config_size=C, embed_size=128, obs_size=6, scale=scale).eval()
    m.load_state_dict(load_weights(name))
    m.mlp_dtype = 'bf16'
    args = [torch.from_numpy(r[k]) for k in ('path', 'free', 'collided', 'edge_index')]
    out = m(path=args[0].to(DEV), free=args[1].to(DEV), collided=args[2].to(DEV), edge_index=args[3].to(DEV),
            loop=int(r['loop'])).cpu()
    emu = ref_bf16.smoother_forward_bf16(load_weights(name), *args, loop=int(r['loop']), scale=scale)
    ref = torch.from_numpy(r['out_fp32'])
    e_emu, e_ref = (out - emu).abs().max().item() / scale, (out - ref).abs().max().item() / scale
    print('\n%s: bf16 gpu vs emulation %.2e, vs fp32 reference %.2e (units of scale)' % (os.path.basename(path), e_emu, e_ref))
    assert e_emu <= 1e-3            # measured 1e-7 .. 1.4e-4: short pipeline, few rounding flips
    assert e_ref <= 2e-2            # measured 1.6e-3 .. 7.9e-3
    assert torch.equal(out[0], ref[0]) and torch.equal(out[-1], ref[-1])          # end points pass through unchanged

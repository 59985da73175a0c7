"""The smoother's tile-per-wave kernels (GNNMP_SM_SPLIT=0) against its tile-per-workgroup kernels (default for fp32): same
bits.  The switch is read once per process, so the forced run happens in a subprocess."""
import os
import subprocess
import sys

import pytest
import torch

from conftest import REPO

pytestmark = pytest.mark.gpu

SCRIPT = r'''
import sys, torch
sys.path.insert(0, %r); sys.path.insert(0, %r + '/tests')
import gnnmp
from gnnmp.weights import load_weights
from gnnmp.planner import chain_edge_index
m = gnnmp.ModelSmoother(workspace_size=3, config_size=7, embed_size=128, obs_size=6).eval()
m.load_state_dict(load_weights('smooth_7d_attv3'))
m.mlp_dtype = sys.argv[1]
gen = torch.Generator().manual_seed(23)
probs = [(torch.rand(20 + 4 * (i %% 4), 7, generator=gen) * 2 - 1, torch.rand(120, 7, generator=gen) * 2 - 1,
          torch.rand(90, 7, generator=gen) * 2 - 1) for i in range(24)]
sb = gnnmp.SmoothBatch([p[0] for p in probs], [p[1] for p in probs], [p[2] for p in probs],
                       [chain_edge_index(p[0].shape[0]) for p in probs], 'cuda:0')
torch.save(m.forward_batch(sb, 2).cpu(), sys.argv[2])
'''


@pytest.mark.parametrize('mode', ['fp32', 'bf16'])
def test_wave_kernels_equal_split_kernels_bitwise(mode, tmp_path):
    outs = []
    for split in ('0', '1'):
        out = str(tmp_path / ('out_%s.pt' % split))
        env = dict(os.environ, GNNMP_SM_SPLIT=split)
        subprocess.run([sys.executable, '-c', SCRIPT % (REPO, REPO), mode, out], check=True, env=env, timeout=300)
        outs.append(torch.load(out))
    assert torch.equal(outs[0], outs[1])
    assert bool(torch.isfinite(outs[0]).all())


STREAM_SCRIPT = r'''
import sys, torch
sys.path.insert(0, %r); sys.path.insert(0, %r + '/tests')
import gnnmp
from gnnmp.weights import load_weights
from gnnmp.planner import chain_edge_index
ckpt, C, n_prob = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
m = gnnmp.ModelSmoother(workspace_size=3, config_size=C, embed_size=128, obs_size=6).eval()
m.load_state_dict(load_weights(ckpt))
gen = torch.Generator().manual_seed(31)
probs = [(torch.rand(3 + (7 * i) %% 38, C, generator=gen) * 2 - 1, torch.rand(40 + (13 * i) %% 300, C, generator=gen) * 2 - 1,
          torch.rand(1 + (11 * i) %% 200, C, generator=gen) * 2 - 1) for i in range(n_prob)]
sb = gnnmp.SmoothBatch([p[0] for p in probs], [p[1] for p in probs], [p[2] for p in probs],
                       [chain_edge_index(p[0].shape[0]) for p in probs], 'cuda:0')
outs = [m.forward_batch(sb, loop).cpu() for loop in (1, 3)]
m.check_status()
torch.save(outs, sys.argv[4])
'''


@pytest.mark.parametrize('switch', ['GNNMP_SM_STREAM'])
@pytest.mark.parametrize('ckpt,C,n_prob', [('smooth_14d_attv3', 14, 256), ('smooth_2d_attv3', 2, 37), ('smooth_7d_attv3', 7, 700)])
def test_streamed_weights_message_kernel_equals_split_kernel_bitwise(ckpt, C, n_prob, switch, tmp_path):
    """sm_msg_stream_kernel (large fp32 batches at d = 128: eight waves, a tile per wave, every matrix through LDS in columns, target
    rows published by rounds of four path tiles) against the split kernel: same bits, ragged problems (3-40 waypoints: one or two
    path tiles each), one and three iterations (the flags are re-armed by every graph stage), batch sizes on both sides of a round
    boundary.  The switch is read once per process, so each form runs in its own subprocess."""
    outs = []
    for stream in ('0', '1'):
        out = str(tmp_path / ('stream_%s.pt' % stream))
        env = dict(os.environ)
        env[switch] = stream
        subprocess.run([sys.executable, '-c', STREAM_SCRIPT % (REPO, REPO), ckpt, str(C), str(n_prob), out], check=True, env=env, timeout=600)
        outs.append(torch.load(out))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
        assert bool(torch.isfinite(a).all())

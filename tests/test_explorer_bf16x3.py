"""'bf16x3' mode of the explorer: every fp32 MFMA operand is split exactly into three bf16 pieces and the six
leading piece products are accumulated in fp32 on the bf16 matrix pipe (chain.hpp, Prec<2>).  The results are
fp32-class, so this mode is held to EXACTLY the same bar as the exact-fp32 kernels (tests/test_explorer_parity.py):
allclose(rtol=1e-5, atol=2e-5) against the fp32 reference goldens and an error against the fp64 run no worse than
2x the reference's own fp32 error."""
import os

import numpy as np
import pytest
import torch

from conftest import env_of, golden_files, load_weights
import gnnmp
from gnnmp.synth import ENVS, synth_graph
from oracle import ref_cpu

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
RTOL, ATOL = 1e-5, 2e-5


def make(env, use_obstacles=True):
    e = ENVS[env]
    m = gnnmp.EncoderProcessDecoder(e['workspace'], e['C'], e['d'], e['S'], use_obstacles=use_obstacles).eval()
    m.load_state_dict(load_weights(e['ckpt']))
    m.mlp_dtype = 'bf16x3'
    return m


@pytest.mark.parametrize('path', golden_files('explorer_'), ids=os.path.basename)
def test_golden_scores_bf16x3(path):
    with np.load(path) as f:
        r = {k: f[k] for k in f.files}
    m = make(env_of(path), bool(r['use_obstacles']))
    s = m.edge_scores(torch.from_numpy(r['goal']).to(DEV), int(r['loop']), torch.from_numpy(r['v']).to(DEV),
                      torch.from_numpy(r['obstacles']).to(DEV), torch.from_numpy(r['edge_index']).to(DEV)).cpu()
    ref32, ref64 = torch.from_numpy(r['scores_fp32']), torch.from_numpy(r['scores_fp64'])
    err32 = (s - ref32).abs().max().item()
    err64 = (s.double() - ref64).abs().max().item()
    own = (ref32.double() - ref64).abs().max().item()
    print('\n%s [bf16x3]: max|gpu-ref32|=%.2e  max|gpu-ref64|=%.2e  (reference fp32-vs-fp64: %.2e)' %
          (os.path.basename(path), err32, err64, own))
    assert torch.allclose(s, ref32, rtol=RTOL, atol=ATOL), err32
    assert err64 <= max(2.0 * own, 2e-5), (err64, own)


@pytest.mark.parametrize('n_obs', [0, 33, 128, 300])
def test_obstacle_counts_bf16x3(n_obs):
    gen = torch.Generator().manual_seed(n_obs)
    v = torch.rand(70, 2, generator=gen) * 2 - 1
    obstacles = torch.rand(n_obs, 2, generator=gen) - 0.5
    ei = ref_cpu.build_edges(v, 35, 4)
    m = make('maze2')
    s = m.edge_scores(v[1].to(DEV), 5, v.to(DEV), obstacles.to(DEV), ei.to(DEV)).cpu()
    ref = ref_cpu.explorer_forward(load_weights('weights_maze'), v, v[1].clone(), obstacles, ei, 5)
    assert torch.allclose(s, ref, rtol=RTOL, atol=ATOL), (s - ref).abs().max()


def test_batch_equals_single_bf16x3():
    graphs = [synth_graph('maze2', n, 5, seed=50 + i, n_obs=o) for i, (n, o) in enumerate(((64, 116), (200, 57), (129, 128)))]
    m = make('maze2')
    b = gnnmp.GraphBatch.from_graphs(graphs, 2, DEV)
    sb = m.forward_batch(b, 4)
    for g, part in zip(graphs, b.split_edges(sb)):
        s1 = m.edge_scores(g['goal'].to(DEV), 4, g['v'].to(DEV), g['obstacles'].to(DEV), g['edge_index'].to(DEV))
        assert torch.equal(s1, part)

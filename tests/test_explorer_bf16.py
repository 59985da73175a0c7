"""bf16 operand mode of the explorer (BASELINE configs[2] and [4]: "bf16 MLP MFMA").

Two bars:
 (i)  the kernels do what the mode claims: GPU vs the CPU emulation of the same formulation with the same
      bf16 rounding points (oracle/ref_bf16.py): mean|d| <= 1e-2 and max|d| <= 0.15 on scores spanning tens
      of units.  The two cannot agree more tightly: their pre-rounding fp32 values differ in the last bits
      (summation order), which flips a bf16 rounding in roughly one operand per 10^4, and every flip is a
      2^-8 relative step that then travels through five message-passing iterations (measured: mean 2-6e-3).
 (ii) accuracy against the fp32 reference goldens on the kuka checkpoints (the bf16 configs): mean|d| <= 0.025,
      max|d| <= 0.15 (the survey's CPU probe of this scheme: 0.010 / 0.050-0.080), and the per-target best
      incoming edge agrees wherever the fp32 top-2 margin exceeds 0.2; overall agreement >= 96 %.
      (The maze checkpoint is NOT a bf16 config: its scores span [-32, 12] and bf16 costs up to ~1 unit
      there; it is exercised for (i) only.)
"""
import os

import numpy as np
import pytest
import torch

from conftest import env_of, golden_files, load_weights
import gnnmp
from gnnmp.synth import ENVS, synth_graph
from oracle import ref_bf16, ref_cpu

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def make(env):
    e = ENVS[env]
    m = gnnmp.EncoderProcessDecoder(e['workspace'], e['C'], e['d'], e['S']).eval()
    m.load_state_dict(load_weights(e['ckpt']))
    m.mlp_dtype = 'bf16'
    return m


@pytest.mark.parametrize('env,n,k', [('maze2', 200, 6), ('kuka7', 300, 8), ('kuka14', 250, 9), ('ur5', 150, 5),
                                     ('snake7', 120, 5)])
def test_matches_bf16_emulation(env, n, k):
    g = synth_graph(env, n, k, seed=77)
    m = make(env)
    s = m.edge_scores(g['goal'].to(DEV), 5, g['v'].to(DEV), g['obstacles'].to(DEV), g['edge_index'].to(DEV)).cpu()
    emu = ref_bf16.explorer_forward_bf16(load_weights(ENVS[env]['ckpt']), g['v'], g['goal'], g['obstacles'], g['edge_index'], 5)
    d = (s - emu).abs()
    print('\n%s: gpu-bf16 vs emulation: max %.2e mean %.2e (score range [%.1f, %.1f])' % (env, d.max(), d.mean(), emu.min(), emu.max()))
    assert float(d.max()) <= 0.15 and float(d.mean()) <= 1e-2


@pytest.mark.parametrize('path', [p for p in golden_files('explorer_kuka')], ids=os.path.basename)
def test_accuracy_vs_fp32_reference(path):
    with np.load(path) as f:
        r = {k: f[k] for k in f.files}
    env = env_of(path)
    m = make(env)
    m.use_obstacles = bool(r['use_obstacles'])
    s = m.edge_scores(torch.from_numpy(r['goal']).to(DEV), int(r['loop']), torch.from_numpy(r['v']).to(DEV),
                      torch.from_numpy(r['obstacles']).to(DEV), torch.from_numpy(r['edge_index']).to(DEV)).cpu()
    ref = torch.from_numpy(r['scores_fp32'])
    d = (s - ref).abs()
    ei = torch.from_numpy(r['edge_index'])
    agree = tot = 0
    for t in ei[1].unique().tolist():
        sel = (ei[1] == t).nonzero().squeeze(1)
        if sel.numel() > 1:
            tot += 1
            same = int(s[sel].argmax()) == int(ref[sel].argmax())
            agree += int(same)
            top = ref[sel].topk(2).values
            if float(top[0] - top[1]) > 0.2:
                assert same, 'argmax flipped across a margin of %.3f' % float(top[0] - top[1])
    print('\n%s: bf16 vs fp32 reference: max %.3f mean %.4f argmax agreement %.2f %%' %
          (os.path.basename(path), d.max(), d.mean(), 100.0 * agree / max(tot, 1)))
    assert float(d.mean()) <= 0.025 and float(d.max()) <= 0.15       # measured 0.0071-0.0209 / 0.045-0.115
    assert agree >= 0.96 * tot


def test_batch_equals_single_and_modes_coexist():
    graphs = [synth_graph('kuka7', n, 6, seed=5 + i) for i, n in enumerate((90, 160, 64))]
    m = make('kuka7')
    b = gnnmp.GraphBatch.from_graphs(graphs, 6, DEV)
    sb = m.forward_batch(b, 3)
    for g, part in zip(graphs, b.split_edges(sb)):
        s1 = m.edge_scores(g['goal'].to(DEV), 3, g['v'].to(DEV), g['obstacles'].to(DEV), g['edge_index'].to(DEV))
        assert torch.equal(s1, part)
    # switching back to fp32 on the same module restores exact-precision results
    m.mlp_dtype = 'fp32'
    g = graphs[0]
    s32 = m.edge_scores(g['goal'].to(DEV), 3, g['v'].to(DEV), g['obstacles'].to(DEV), g['edge_index'].to(DEV)).cpu()
    ref = ref_cpu.explorer_forward(load_weights('weights_kuka'), g['v'], g['goal'], g['obstacles'], g['edge_index'], 3)
    assert torch.allclose(s32, ref, rtol=1e-5, atol=2e-5)


# max|gpu_bf16 - reference_fp64| per golden, recorded on the MI355X in round 3 (profiles/r03_parity.txt section G).  The emulation
# oracle (ref_bf16.py) follows THIS implementation's rounding points, so agreement with it cannot pin the mode; this budget
# against the unmodified reference's fp64 run does: an algebraic refold that shifts bf16 accuracy shows up here.
BF16_BUDGET = {
    'explorer_kuka13_N64_k4_L5': 7.621e-02, 'explorer_kuka14_N200_k8_L5': 8.443e-02, 'explorer_kuka14_N64_k4_L5': 1.018e-01,
    'explorer_kuka7_N200_k6_L5': 5.678e-02, 'explorer_kuka7_N64_k4_L2_noobs': 4.986e-02, 'explorer_kuka7_N64_k4_L5': 4.704e-02,
    'explorer_snake7_N64_k4_L5': 7.292e-02, 'explorer_maze3_N64_k4_L5': 1.211e-01, 'explorer_ur5_N64_k4_L5': 2.673e-01,
    'explorer_maze2_N64_k4_L5_noobs': 3.514e-02,
}


@pytest.mark.parametrize('name', sorted(BF16_BUDGET))
def test_bf16_error_budget(name):
    path = [p for p in golden_files('explorer_') if os.path.basename(p) == name + '.npz'][0]
    with np.load(path) as f:
        r = {k: f[k] for k in f.files}
    m = make(env_of(path))
    m.use_obstacles = bool(r['use_obstacles'])
    s = m.edge_scores(torch.from_numpy(r['goal']).to(DEV), int(r['loop']), torch.from_numpy(r['v']).to(DEV),
                      torch.from_numpy(r['obstacles']).to(DEV), torch.from_numpy(r['edge_index']).to(DEV)).cpu().double()
    err = float((s - torch.from_numpy(r['scores_fp64'])).abs().max())
    print('\n%s: bf16 vs reference fp64 max %.3e (recorded %.3e)' % (name, err, BF16_BUDGET[name]))
    assert err <= 1.3 * BF16_BUDGET[name]

"""bf16 operand mode against an EXTERNALLY produced yardstick: the unmodified reference module cast to bfloat16.

tests/golden/refbf16_*.npz hold what ``/root/reference/model.py``'s ``EncoderProcessDecoder`` returns after
``module.to(torch.bfloat16)`` on the inputs of the kuka explorer goldens and on one full-size graph of each bf16 BASELINE
shape (configs[2]: kuka7 2000-node k=10, configs[4]: kuka14 5000-node k=16), recorded by ``tools/gen_golden.py bf16anchor``
next to the reference's fp32 scores.  It is the only bf16 run the reference can make (``torch.autocast`` stops at
model.py:134, and the smoother concatenates a float32 one-hot block, model_smoother.py:131-135, so it has no bf16 run
at all).  ``mlp_dtype='bf16'`` rounds MFMA operands only (fp32 accumulators, LayerNorm, softmax statistics, aggregation),
so the bar is: at least as close to the reference's fp32 scores as the reference's own bf16 run -- in the mean, in the
maximum, and in the decisions (per-target best incoming edge, the quantity eval_gnn.py:205-211 consumes).
oracle/ref_bf16.py (the CPU emulation of the kernels' rounding points) is held to the same yardstick on the CPU.
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, golden_files, load_weights
import gnnmp
from gnnmp.synth import ENVS, synth_graph
from oracle import ref_bf16

ANCHORS = golden_files('refbf16_explorer_')
FULL = golden_files('refbf16_full_')


def _load(path):
    with np.load(path) as f:
        return {k: f[k] for k in f.files}


def _agreement(scores, ref, ei):
    """share of targets with >= 2 incoming edges whose best incoming edge is the fp32 reference's"""
    order = torch.argsort(ei[1], stable=True)
    t = ei[1][order]
    starts = torch.cat((torch.tensor([0]), (t[1:] != t[:-1]).nonzero().squeeze(1) + 1, torch.tensor([t.numel()])))
    agree = tot = 0
    for a, b in zip(starts[:-1].tolist(), starts[1:].tolist()):
        if b - a > 1:
            sel = order[a:b]
            tot += 1
            agree += int(int(scores[sel].argmax()) == int(ref[sel].argmax()))
    return agree / max(tot, 1), tot


def _case(path):
    a = _load(path)
    r = _load(os.path.join(GOLDEN, str(a['of'])))
    env = str(a['of']).split('_')[1]
    return env, r, torch.from_numpy(a['scores_ref_bf16'])


def _check(name, got, ref32, ref_bf16, ei):
    d, da = (got - ref32).abs(), (ref_bf16 - ref32).abs()
    (ag, tot), (aga, _) = _agreement(got, ref32, ei), _agreement(ref_bf16, ref32, ei)
    print('\n%s: |ours - ref32| max %.4f mean %.4f agree %.2f %%   |reference-in-bf16 - ref32| max %.4f mean %.4f agree %.2f %%'
          % (name, d.max(), d.mean(), 100 * ag, da.max(), da.mean(), 100 * aga))
    assert float(d.mean()) <= float(da.mean()), name
    assert float(d.max()) <= float(da.max()), name
    # decisions: a flip needs a fp32 top-2 margin below the two scores' errors, so on 64 targets the count is 0-2 either way;
    # the bar is the reference-in-bf16's agreement less 1 % (at least three targets: measured differences are 0-2 targets either way)
    assert ag >= aga - max(0.01, 3.0 / tot), name


@pytest.mark.parametrize('path', ANCHORS, ids=os.path.basename)
def test_emulation_is_at_least_as_accurate_as_the_reference_in_bf16(path):
    """CPU: the emulation of the kernels' bf16 mode against the reference's own bf16 run."""
    env, r, anchor = _case(path)
    w = load_weights(ENVS[env]['ckpt'])
    emu = ref_bf16.explorer_forward_bf16(w, torch.from_numpy(r['v']), torch.from_numpy(r['goal']), torch.from_numpy(r['obstacles']),
                                         torch.from_numpy(r['edge_index']), int(r['loop']), use_obstacles=bool(r['use_obstacles']))
    _check(os.path.basename(path), emu, torch.from_numpy(r['scores_fp32']), anchor, torch.from_numpy(r['edge_index']))


def _model(env):
    e = ENVS[env]
    m = gnnmp.EncoderProcessDecoder(e['workspace'], e['C'], e['d'], e['S']).eval()
    m.load_state_dict(load_weights(e['ckpt']))
    m.mlp_dtype = 'bf16'
    return m


@pytest.mark.gpu
@pytest.mark.parametrize('path', ANCHORS, ids=os.path.basename)
def test_kernels_are_at_least_as_accurate_as_the_reference_in_bf16(path):
    env, r, anchor = _case(path)
    m = _model(env)
    m.use_obstacles = bool(r['use_obstacles'])
    dev = 'cuda:0'
    s = m.edge_scores(torch.from_numpy(r['goal']).to(dev), int(r['loop']), torch.from_numpy(r['v']).to(dev),
                      torch.from_numpy(r['obstacles']).to(dev), torch.from_numpy(r['edge_index']).to(dev)).cpu()
    _check(os.path.basename(path), s, torch.from_numpy(r['scores_fp32']), anchor, torch.from_numpy(r['edge_index']))


@pytest.mark.gpu
@pytest.mark.parametrize('path', FULL, ids=os.path.basename)
def test_full_size_bf16_shapes_against_the_reference_in_bf16(path):
    """BASELINE configs[2] / configs[4] shapes at full size: the graph is regenerated from its seed (checked), scored alone and
    inside a ragged batch (same bits), and held to the recorded reference runs."""
    a = _load(path)
    env, n, k, seed = str(a['env']), int(a['n']), int(a['k']), int(a['seed'])
    g = synth_graph(env, n, k, seed=seed)
    assert g['edge_index'].shape[1] == int(a['n_edges']) and int(g['edge_index'].long().sum()) == int(a['ei_sum'])
    assert float(g['v'].double().sum()) == float(a['v_sum'])
    m = _model(env)
    dev = 'cuda:0'
    s = m.edge_scores(g['goal'].to(dev), int(a['loop']), g['v'].to(dev), g['obstacles'].to(dev), g['edge_index'].to(dev)).cpu()
    _check(os.path.basename(path), s, torch.from_numpy(a['scores_fp32']), torch.from_numpy(a['scores_ref_bf16']), g['edge_index'])
    others = [synth_graph(env, n // 2 + 17 * i, k, seed=seed + 1 + i) for i in range(2)]
    batch = gnnmp.GraphBatch.from_graphs([others[0], g, others[1]], ENVS[env]['S'], dev)
    parts = batch.split_edges(m.forward_batch(batch, int(a['loop'])))
    assert torch.equal(parts[1].cpu(), s)


def test_full_size_yardstick_belongs_to_the_seeded_graphs():
    """CPU: tests/golden/refbf16_stats_full.npz was recorded on synth_graph(env, n, k, seed) -- the oracle's materialising fp32 form
    (bit-identical to the reference's fp32 run) reproduces the recorded sum of the reference's fp32 scores for the first kuka7 graph."""
    from oracle import ref_cpu
    with np.load(os.path.join(GOLDEN, 'refbf16_stats_full.npz')) as f:
        st = {k: f[k] for k in f.files}
    assert len(st['seed']) == 17 and float(st['err_mean'].max()) < 0.06 and float(st['err_max'].max()) < 0.5
    i = int(np.nonzero((st['env'] == 'kuka7') & (st['seed'] == 1234))[0][0])
    g = synth_graph('kuka7', int(st['n'][i]), int(st['k'][i]), seed=1234)
    s = ref_cpu.explorer_forward(load_weights(ENVS['kuka7']['ckpt']), g['v'], g['goal'], g['obstacles'], g['edge_index'], 5, materialize=True)
    assert float(s.double().sum()) == float(st['ref32_sum'][i])

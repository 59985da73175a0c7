"""BASELINE configs[2] and [4] at their REAL shapes on one GPU, in both operand modes, through size-independent
properties (the style of tests/test_full_size_gpu.py):

  cfg 3 (configs[2]): kuka7 / weights_kuka (C 7, d 64, S 6), 64 problems x 2000-node k1=10 RGGs (E ~ 30.7 k, in-degree
                      10-33: rows span several 32-edge tiles), O = 5;
  cfg 5 (configs[4]): kuka14 / kuka_14 (C 14, d 32), 32 problems x 5000-node k1=16 RGGs (E ~ 131 k, in-degree 16-90)
                      + the smooth_14d_attv3 smoother (d 128) on batches of planning paths.

 (a) a graph scored inside the batch == the same graph scored alone, bit for bit (fp32 and bf16);
 (b) two runs give identical bytes;
 (c) sampled graphs: fp32 mode against the fp32 / fp64 oracle with the per-fixture bar of tests/parity_bar.py;
     bf16 mode against the CPU emulation of the same rounding points (oracle/ref_bf16.py) and against the fp32 oracle
     with the reference's own bf16 run of the same graph as the yardstick (tests/golden/refbf16_stats_full.npz);
 (d) device-built graphs == host builder for the sampled seeds (so the oracle sees the same inputs)."""
import math
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_weights
import gnnmp
from gnnmp.synth import ENVS, synth_batch_gpu, synth_graph
from oracle import ref_bf16
from parity_bar import assert_fp32_parity, explorer_oracle_pair

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _model(env, dtype):
    e = ENVS[env]
    m = gnnmp.EncoderProcessDecoder(e['workspace'], e['C'], e['d'], e['S']).eval()
    m.load_state_dict(load_weights(e['ckpt']), strict=True)
    m.mlp_dtype = dtype
    return m


def _yardstick(env, n, k, seed):
    """(max, mean) of |reference in bf16 - reference in fp32| on synth_graph(env, n, k, seed): the recorded reference runs"""
    with np.load(os.path.join(GOLDEN, 'refbf16_stats_full.npz')) as f:
        sel = (f['env'] == env) & (f['n'] == n) & (f['k'] == k) & (f['seed'] == seed)
        assert int(sel.sum()) == 1, (env, n, k, seed)
        return float(f['err_max'][sel][0]), float(f['err_mean'][sel][0])


def _argmax_agreement(s, ref, ei, margin):
    agree = tot = 0
    order = torch.argsort(ei[1], stable=True)
    t_sorted = ei[1][order]
    bounds = torch.nonzero(torch.cat((torch.tensor([True]), t_sorted[1:] != t_sorted[:-1], torch.tensor([True])))).view(-1).tolist()
    for a, b in zip(bounds[:-1], bounds[1:]):
        if b - a < 2:
            continue
        sel = order[a:b]
        tot += 1
        same = int(s[sel].argmax()) == int(ref[sel].argmax())
        agree += int(same)
        top = ref[sel].topk(2).values
        if float(top[0] - top[1]) > margin:
            assert same, 'argmax flipped across a margin of %.3f' % float(top[0] - top[1])
    return agree, tot


@pytest.mark.parametrize('env,G,N,K1,sample', [('kuka7', 64, 2000, 10, (0, 41)), ('kuka14', 32, 5000, 16, (1, 3))],
                         ids=['cfg3_kuka7_2000_k10_x64', 'cfg5_kuka14_5000_k16_x32'])
def test_full_size_both_modes(env, G, N, K1, sample):
    e = ENVS[env]
    graphs = synth_batch_gpu(env, N, K1, G, DEV)
    b = gnnmp.GraphBatch.from_graphs(graphs, e['S'], DEV)
    w = load_weights(e['ckpt'])
    host = {}
    for i in sample:
        host[i] = synth_graph(env, N, K1, seed=1234 + i)
        assert torch.equal(graphs[i]['edge_index'].cpu(), host[i]['edge_index'])          # (d)
        assert torch.equal(graphs[i]['v'].cpu(), host[i]['v'])
    deg = torch.bincount(host[sample[0]]['edge_index'][1], minlength=N)
    print('\n%s: E per graph %d, in-degree %d-%d' % (env, host[sample[0]]['edge_index'].shape[1], int(deg.min()), int(deg.max())))
    assert int(deg.max()) > 32                                   # rows that span several 32-edge tiles are present
    for dtype in ('fp32', 'bf16'):
        ratios = []
        m = _model(env, dtype)
        s1 = m.forward_batch(b, 5).clone()
        s2 = m.forward_batch(b, 5)
        assert torch.equal(s1, s2), dtype                                                  # (b)
        assert bool(torch.isfinite(s1).all())
        parts = b.split_edges(s1)
        for i in sample:
            g = graphs[i]
            alone = m.edge_scores(g['goal'], 5, g['v'], g['obstacles'], g['edge_index'])
            assert torch.equal(alone, parts[i]), (dtype, i)                                # (a)
            got = parts[i].cpu()
            ref32, ref64 = explorer_oracle_pair(w, host[i], 5)
            if dtype == 'fp32':                                                            # (c)
                c = assert_fp32_parity(got, ref32, ref64, '%s graph %d' % (env, i))
                print('%s graph %d fp32: |gpu-ref32| %.2e |gpu-ref64| %.2e own %.2e bar %.2e' %
                      (env, i, c['err32'], c['err64'], c['own'], c['atol']))
            else:
                emu = ref_bf16.explorer_forward_bf16(w, host[i]['v'], host[i]['goal'], host[i]['obstacles'],
                                                     host[i]['edge_index'], 5)
                d_emu, d_ref = (got - emu).abs(), (got - ref32).abs()
                agree, tot = _argmax_agreement(got, ref32, host[i]['edge_index'], 0.2)
                print('%s graph %d bf16: vs emulation max %.3f mean %.4f | vs fp32 oracle max %.3f mean %.4f | best-incoming-edge '
                      'agreement %.2f %% of %d targets' % (env, i, d_emu.max(), d_emu.mean(), d_ref.max(), d_ref.mean(),
                                                           100.0 * agree / max(tot, 1), tot))
                ya_max, ya_mean = _yardstick(env, N, K1, 1234 + i)
                # regression guard against the emulation: rounding flips between two bf16 pipelines are amplified by the graph exactly
                # like the mode's own error, so on the graphs where the reference-in-bf16 is far off (kuka14 seed 1235: mean 0.047)
                # the two drift further apart as well (measured 0.0137 there, 0.004-0.007 elsewhere)
                assert float(d_emu.max()) <= max(0.15, 0.75 * ya_max) and float(d_emu.mean()) <= max(1e-2, 0.5 * ya_mean)
                # ... and absolute ceilings next to the relative bars (a yardstick that is itself far off must not widen them without
                # bound): the worst sampled graph measures 0.144 / 0.0137 against the emulation and 0.35 / 0.028 against the fp32 oracle
                assert float(d_emu.max()) <= 0.20 and float(d_emu.mean()) <= 0.02
                assert float(d_ref.max()) <= 0.50 and float(d_ref.mean()) <= 0.04
                # accuracy against the fp32 reference, measured with the reference's OWN bf16 run of the same graph as the yardstick
                # (tests/golden/refbf16_stats_full.npz: the unmodified module cast to bfloat16, recorded by tools/gen_golden.py
                # bf16anchor).  Over sixteen full-size graphs the kernels' mean error is 0.50-0.81 x the reference-in-bf16's on
                # fifteen and 1.26 x on one (kuka14 seed 1237, sampled here on purpose); per graph the bar is 1.5 x (mean) and
                # 1.6 x (max) of the yardstick, over the sampled graphs of a shape the geometric mean of the ratio stays <= 1
                print('   reference in bf16 on the same graph: max %.3f mean %.4f  -> ratio max %.2f mean %.2f' %
                      (ya_max, ya_mean, float(d_ref.max()) / ya_max, float(d_ref.mean()) / ya_mean))
                assert float(d_ref.mean()) <= 1.5 * ya_mean and float(d_ref.max()) <= 1.6 * ya_max
                ratios.append(float(d_ref.mean()) / ya_mean)
                assert agree >= 0.96 * tot
        if dtype == 'bf16':
            assert math.exp(sum(math.log(r) for r in ratios) / len(ratios)) <= 1.0, ratios


def test_cfg5_smoother_batch_both_modes():
    """configs[4] "+ smoother GNN": smooth_14d_attv3 on 256 planning paths (P 5-35 waypoints, 500 + 500 samples each,
    the reference's per-call maximum, smoother.py:53-58): batch == single bit for bit, deterministic, fp32 within the
    smoother bar of the oracle, bf16 within the bf16 bar of tests/test_smoother_bf16.py."""
    from gnnmp.planner import chain_edge_index
    from gnnmp.smoother import SmoothBatch
    from oracle import ref_cpu
    C, B = 14, 256
    gen = torch.Generator().manual_seed(14)
    w = load_weights('smooth_14d_attv3')
    lim = torch.tensor(ENVS['kuka14']['lim'])
    mk = lambda n: ((torch.rand(n, C, generator=gen) * 2 - 1) * lim)          # noqa: E731
    Ps = [int(torch.randint(5, 36, (1,), generator=gen)) for _ in range(B)]
    paths, frees, colls = [mk(p) for p in Ps], [mk(500) for _ in range(B)], [mk(500) for _ in range(B)]
    eis = [chain_edge_index(p) for p in Ps]
    sb = SmoothBatch(paths, frees, colls, eis, DEV)
    outs = {}
    for dtype in ('fp32', 'bf16'):
        ms = gnnmp.ModelSmoother(workspace_size=3, config_size=C, embed_size=128, obs_size=6, scale=1.0).eval()
        ms.load_state_dict(w, strict=True)
        ms.mlp_dtype = dtype
        o1 = ms.forward_batch(sb, 1).clone()
        o2 = ms.forward_batch(sb, 1)
        assert torch.equal(o1, o2)
        outs[dtype] = o1.cpu()
        off = 0
        for i, p in enumerate(Ps):
            if i in (0, 100, 255):
                single = ms(path=paths[i].to(DEV), free=frees[i].to(DEV), collided=colls[i].to(DEV), edge_index=eis[i].to(DEV), loop=1)
                assert torch.equal(single.cpu(), outs[dtype][off:off + p]), (dtype, i)
                ref = ref_cpu.smoother_forward(w, paths[i], frees[i], colls[i], eis[i], loop=1, scale=1.0)
                err = (outs[dtype][off:off + p] - ref).abs().max().item()
                if dtype == 'fp32':
                    assert torch.allclose(outs[dtype][off:off + p], ref, rtol=1e-5, atol=1e-5), err
                else:
                    assert err <= 2e-2, err
            off += p
    print('\nsmooth_14d batch of %d: bf16 vs fp32 max %.2e' % (B, (outs['bf16'] - outs['fp32']).abs().max()))


_W8_SNIPPET = r'''
import hashlib, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import torch
from conftest import load_weights
import gnnmp
from gnnmp.synth import ENVS, synth_graph
e = ENVS['kuka7']
sizes = [300 + 41 * (i %% 23) + 7 * (i %% 40) for i in range(%d)]
graphs = [synth_graph('kuka7', n, 6, seed=700 + i) for i, n in enumerate(sizes)]
m = gnnmp.EncoderProcessDecoder(e['workspace'], e['C'], e['d'], e['S']).eval(); m.load_state_dict(load_weights(e['ckpt'])); m.mlp_dtype = %r
b = gnnmp.GraphBatch.from_graphs(graphs, e['S'], 'cuda:0')
for loop in (1, 4):
    s = m.forward_batch(b, loop)
    torch.cuda.synchronize()
    print('HASH', loop, hashlib.sha256(s.cpu().numpy().tobytes()).hexdigest())
'''


@pytest.mark.parametrize('mode', ['bf16', 'fp32'])
@pytest.mark.parametrize('n_graphs', [40, 400], ids=['one_group_per_virtual_workgroup', 'resident_workgroups_on_the_snake'])
def test_d64_eight_wave_kernel_equals_four_wave_kernel(n_graphs, mode):
    """d = 64 (kuka7: BASELINE configs[2] in bf16, a member of configs[3] in fp32) runs mp_fused_w8_kernel on large batches: eight
    waves per CU, a tile's B' rows in registers (ds_bpermute per chunk) instead of an LDS stage and -- bf16 -- every matrix of the
    message layer and the node phase in LDS.  Same arithmetic in the same order as mp_fused_kernel<64, P, 1> (GNNMP_MP_W8=0): the
    scores are the same BYTES in both operand modes -- on a
    ragged batch (graphs of 300 ... 1400 nodes with differing numbers of 256-row blocks) small enough for one group per virtual
    workgroup (40 graphs) and large enough for the resident workgroups to walk the snake (400 graphs, ~2900 groups > 512 virtual
    slots), for loop = 1 (first iteration == last: the staged W_dst copy) and loop = 4, and equal to per-graph calls (which run
    the tile-per-workgroup form)."""
    import hashlib
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = _W8_SNIPPET % (os.path.dirname(here), here, n_graphs, mode)
    got = {}
    for w8, order in (('0', ''), ('1', ''), ('1', '0'), ('1', '2')):
        env = dict(os.environ, GNNMP_MP_W8=w8)
        env.pop('GNNMP_MP_ORDER', None)
        if order:
            env['GNNMP_MP_ORDER'] = order
        out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        got[(w8, order)] = [ln for ln in out.stdout.splitlines() if ln.startswith('HASH')]
    assert len(got[('0', '')]) == 2
    assert got[('0', '')] == got[('1', '')] == got[('1', '0')] == got[('1', '2')], got
    if n_graphs <= 40:
        e = ENVS['kuka7']
        sizes = [300 + 41 * (i % 23) + 7 * (i % 40) for i in range(n_graphs)]
        graphs = [{k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in synth_graph('kuka7', n, 6, seed=700 + i).items()} for i, n in enumerate(sizes)]
        m = _model('kuka7', mode)
        alone = torch.cat([m.edge_scores(g['goal'], 4, g['v'], g['obstacles'], g['edge_index']) for g in graphs])
        assert 'HASH 4 ' + hashlib.sha256(alone.cpu().numpy().tobytes()).hexdigest() == got[('1', '')][1]

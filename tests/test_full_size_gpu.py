"""BASELINE configs[1] at FULL size (256 problems x 1000-node k=8 RGGs on one GPU), checked through
size-independent properties: problems are independent, so (a) any graph scored inside the 256-batch equals
the same graph scored alone, bit for bit; (b) two runs give identical bytes (no order-dependent atomics in
the results); (c) the dense blocks are exactly the scatter of the sparse scores; (d) sampled graphs agree with
the CPU oracle within the fp32 bar; plus the mixed-environment batching helper of configs[3]."""
import pytest
import torch

from conftest import load_weights
import gnnmp
from gnnmp.synth import ENVS, synth_batch_gpu, synth_graph
from oracle import ref_cpu
from parity_bar import assert_fp32_parity, explorer_oracle_pair

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def test_cfg2_full_batch_properties():
    G, N, K1 = 256, 1000, 8
    graphs = synth_batch_gpu('maze2', N, K1, G, DEV)
    m = gnnmp.EncoderProcessDecoder(2, 2, 32, 2).eval()
    m.load_state_dict(load_weights('weights_maze'))
    b = gnnmp.GraphBatch.from_graphs(graphs, 2, DEV)
    s1 = m.forward_batch(b, 5).clone()
    s2 = m.forward_batch(b, 5)
    assert torch.equal(s1, s2)                                           # (b) deterministic
    parts = b.split_edges(s1)
    w = load_weights('weights_maze')
    for i in (0, 97, 255):
        g = graphs[i]
        alone = m.edge_scores(g['goal'], 5, g['v'], g['obstacles'], g['edge_index'])
        assert torch.equal(alone, parts[i])                              # (a) batch == per graph
        # the device-built graph equals the host builder's for the same seed
        host = synth_graph('maze2', N, K1, seed=1234 + i)
        assert torch.equal(g['edge_index'].cpu(), host['edge_index']) and torch.equal(g['v'].cpu(), host['v'])
        # (d) the fp32 bar of tests/test_explorer_parity.py: no worse against the fp64 run than twice the oracle's own
        # fp32 rounding error on this graph, and allclose against the fp32 oracle up to that noise floor
        ref = ref_cpu.explorer_forward(w, host['v'], host['goal'], host['obstacles'], host['edge_index'], 5)
        w64 = {k: (t.double() if t.is_floating_point() else t) for k, t in w.items()}
        ref64 = ref_cpu.explorer_forward(w64, host['v'].double(), host['goal'].double(), host['obstacles'].double(),
                                         host['edge_index'], 5)
        c = assert_fp32_parity(parts[i].cpu(), ref, ref64, 'graph %d' % i)
        print('graph %d: |gpu-ref64| %.2e  |gpu-ref32| %.2e  oracle fp32-vs-fp64 %.2e  bar %.2e' % (i, c['err64'], c['err32'], c['own'], c['atol']))
    sub = gnnmp.GraphBatch.from_graphs(graphs[:8], 2, DEV)
    sc, dense = m.forward_batch(sub, 5, dense=True)
    off = 0
    for g, p in zip(graphs[:8], sub.split_edges(sc)):
        P = dense[off:off + N * N].view(N, N)
        ei = g['edge_index']
        assert torch.equal(P[ei[1], ei[0]], p)                           # (c) dense == scatter of sparse
        assert int((P != 0).sum()) <= ei.shape[1]
        off += N * N


def test_mixed_environment_set():
    """configs[3]: maze / snake / ur5 / kuka problems in one job -- bucketed per environment family (each
    family has its own (C, d, S) and checkpoint), results returned in the caller's problem order."""
    from gnnmp.dist import run_mixed
    order = ['maze2', 'kuka7', 'ur5', 'snake7', 'kuka7', 'maze2', 'ur5']
    problems = [dict(env=env, **{k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in
                                 synth_graph(env, 90 + 10 * i, 5, seed=40 + i).items()}) for i, env in enumerate(order)]
    models = {}
    for env in set(order):
        e = ENVS[env]
        mm = gnnmp.EncoderProcessDecoder(e['workspace'], e['C'], e['d'], e['S']).eval()
        mm.load_state_dict(load_weights(e['ckpt']))
        models[env] = mm
    scores = run_mixed(problems, models, loop=4)
    assert len(scores) == len(problems)
    for p, s in zip(problems, scores):
        g = {k: p[k].cpu() for k in ('v', 'goal', 'obstacles', 'edge_index')}
        ref32, ref64 = explorer_oracle_pair(load_weights(ENVS[p['env']]['ckpt']), g, 4)
        assert_fp32_parity(s.cpu(), ref32, ref64, p['env'])

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


_RAGGED_SNIPPET = r'''
import hashlib, sys, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
from conftest import load_weights
import gnnmp
from gnnmp.synth import synth_graph
sizes = [300 + 37 * (i %% 29) + 5 * (i %% 72) for i in range(216)]     # 300 .. 1700 nodes: neighbours differ in their number of 256-row blocks
graphs = [{k: (v.to('cuda:0') if torch.is_tensor(v) else v) for k, v in synth_graph('maze2', n, 6, seed=900 + i).items()} for i, n in enumerate(sizes)]
m = gnnmp.EncoderProcessDecoder(2, 2, 32, 2).eval(); m.load_state_dict(load_weights('weights_maze')); m.mlp_dtype = %r
b = gnnmp.GraphBatch.from_graphs(graphs, 2, 'cuda:0')
s = m.forward_batch(b, 5)
torch.cuda.synchronize()
print('HASH', hashlib.sha256(s.cpu().numpy().tobytes()).hexdigest(), int(b.total_nodes))
'''


@pytest.mark.parametrize('mode', ['fp32', 'bf16'])
def test_ragged_large_batch_dispatch_orders(mode):
    """The message-passing launch walks the four-tile groups of a large batch in one of three orders (plain, mirrored pairs per
    workgroup, or resident workgroups on a snake over the XCD's blocks: csrc/explorer_kernels.hip mp_fused_kernel): on a RAGGED
    batch -- 216 graphs of 300 ... 1700 nodes with differing numbers of 256-row blocks, ~1700 groups, i.e. more than the resident
    workgroup slots, so the snake takes several turns -- all give the bytes of the per-graph calls (which run the
    tile-per-workgroup form), i.e. every group is visited exactly once whatever the order."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = _RAGGED_SNIPPET % (os.path.dirname(here), here, mode)
    hashes = {}
    for order in ('0', '1', '2', ''):
        env = dict(os.environ)
        env.pop('GNNMP_MP_ORDER', None)
        if order:
            env['GNNMP_MP_ORDER'] = order
        out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        hashes[order or 'auto'] = [ln for ln in out.stdout.splitlines() if ln.startswith('HASH')][0].split()[1]
    assert len(set(hashes.values())) == 1, hashes
    # and against per-graph calls in this process
    import hashlib
    sizes = [300 + 37 * (i % 29) + 5 * (i % 72) for i in range(216)]
    graphs = [{k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in synth_graph('maze2', n, 6, seed=900 + i).items()} for i, n in enumerate(sizes)]
    m = gnnmp.EncoderProcessDecoder(2, 2, 32, 2).eval()
    m.load_state_dict(load_weights('weights_maze'))
    m.mlp_dtype = mode
    alone = torch.cat([m.edge_scores(g['goal'], 5, g['v'], g['obstacles'], g['edge_index']) for g in graphs])
    assert hashlib.sha256(alone.cpu().numpy().tobytes()).hexdigest() == hashes['auto']

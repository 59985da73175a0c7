"""Training path of the explorer: gradients of the HIP backward against torch.autograd
through the CPU oracle with the reference's detach points (model.py:141,142,146), fp64-anchored:

    |g_gpu - g_oracle64| <= max(1e-4 * max|g_oracle64|, 4 * own) + 1e-6    per parameter tensor,
    own = max|g_oracle32 - g_oracle64|: the same oracle run in fp32 (fp32 sums of ~10^3-10^4 terms with float atomics;
    relu masks and the arg-max of the max-aggregation react to last-digit differences of the forward, so the fp32
    oracle's own deviation is the natural unit, as for the forward bar)

on the 64-node goldens of every checkpoint family, for a random linear loss and for the reference's loss form
(-log_softmax over a frontier row set, train_explorer.py:174); frozen parameters get no gradient; the forward of the
training path equals the inference forward."""
import os

import numpy as np
import pytest
import torch

from conftest import env_of, golden_files, load_weights
import gnnmp
from gnnmp.explorer import TRAINABLE
from gnnmp.synth import ENVS
from oracle import ref_cpu

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _oracle_grads(w, r, loop, loss_fn, dtype):
    wd = {k: (t.to(dtype).clone().requires_grad_(True) if t.is_floating_point() else t) for k, t in w.items()}
    s = ref_cpu.explorer_forward(wd, torch.from_numpy(r['v']).to(dtype), torch.from_numpy(r['goal']).to(dtype),
                                 torch.from_numpy(r['obstacles']).to(dtype), torch.from_numpy(r['edge_index']), loop,
                                 use_obstacles=bool(r['use_obstacles']), detach=True)
    loss_fn(s).backward()
    return s.detach(), {k: t.grad for k, t in wd.items() if torch.is_tensor(t) and t.is_floating_point()}


@pytest.mark.parametrize('path', [p for p in golden_files('explorer_') if 'N64' in p], ids=os.path.basename)
def test_gradients_match_oracle(path):
    with np.load(path) as f:
        r = {k: f[k] for k in f.files}
    env = env_of(path)
    e = ENVS[env]
    w = load_weights(e['ckpt'])
    L = int(r['loop'])
    m = gnnmp.EncoderProcessDecoder(e['workspace'], e['C'], e['d'], e['S'], use_obstacles=bool(r['use_obstacles']))
    m.load_state_dict(w, strict=True)
    m.train()
    g = dict(goal=torch.from_numpy(r['goal']).to(DEV), v=torch.from_numpy(r['v']).to(DEV),
             obstacles=torch.from_numpy(r['obstacles']).to(DEV), edge_index=torch.from_numpy(r['edge_index']).to(DEV))
    b = m._single(g['goal'], g['v'], g['obstacles'], g['edge_index'])
    E = r['edge_index'].shape[1]
    gen = torch.Generator().manual_seed(E)
    coef = torch.randn(E, generator=gen, dtype=torch.float64)
    ei = torch.from_numpy(r['edge_index'])
    rows = ei[1] < 8                                                      # "frontier": edges into the first eight nodes
    pick = int(rows.nonzero()[3])

    def loss_lin(s):
        return (s * coef.to(s.dtype).to(s.device)).sum()

    def loss_ce(s):                                                       # train_explorer.py:174
        return -s[rows.to(s.device)].log_softmax(dim=0)[int(rows[:pick].sum())]

    for name, loss_fn in (('linear', loss_lin), ('cross-entropy', loss_ce)):
        m.zero_grad()
        s = m.train_scores(b, L)
        s_inf = m.forward_batch(b, L)
        assert torch.allclose(s.detach(), s_inf, rtol=1e-5, atol=2e-5)      # same function as the inference path
        loss_fn(s).backward()
        _, g64 = _oracle_grads(w, r, L, loss_fn, torch.float64)
        _, g32 = _oracle_grads(w, r, L, loss_fn, torch.float32)
        worst = 0.0
        for pname, p in m.named_parameters():
            top = pname.split('.')[0]
            ref = g64.get(pname)
            if top not in TRAINABLE or ref is None or pname not in dict(m._manifest):
                if top not in TRAINABLE:
                    assert p.grad is None or float(p.grad.abs().max()) == 0.0, pname     # behind the reference's detach
                continue
            assert p.grad is not None, pname
            scale = float(ref.abs().max())
            err = float((p.grad.cpu().double() - ref).abs().max())
            own = float((g32[pname].double() - ref).abs().max())
            worst = max(worst, err / (scale + 1e-30))
            assert err <= max(1e-4 * scale, 4.0 * own) + 1e-6, (name, pname, err, scale, own)
        print('\n%s, %s loss: worst relative gradient error %.2e' % (os.path.basename(path), name, worst))


def test_dense_training_call_and_optimizer_step():
    """The reference's training call shape (dense policy matrix, indexing + log_softmax, optimizer step,
    train_explorer.py:156-186): the loss goes down on a fixed problem, frozen parameters do not move."""
    from gnnmp.synth import synth_graph
    g = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in synth_graph('maze2', 120, 5, seed=12).items()}
    m = gnnmp.EncoderProcessDecoder(2, 2, 32, 2)
    m.load_state_dict(load_weights('weights_maze'))
    m.to(DEV)
    m.train()
    frozen_before = m.edge_attentions[0].attention.key.weight.detach().clone()
    opt = torch.optim.Adam([p for n, p in m.named_parameters() if n.split('.')[0] in TRAINABLE], lr=1e-3)
    frontier = torch.tensor([0, 3, 7], device=DEV)
    losses = []
    for _ in range(6):
        opt.zero_grad()
        # the reference's own call: the module itself, in train() mode, extra Data fields passed through (train_explorer.py:156-160)
        P = m(goal=g['goal'], loop=3, v=g['v'], obstacles=g['obstacles'], free=g['v'][:60], collided=g['v'][60:],
              edge_index=g['edge_index'], labels=torch.zeros(120, 3, device=DEV), k=10)
        assert P.grad_fn is not None and P.shape == (120, 120)
        cand = P[frontier].reshape(-1)
        loss = -cand.log_softmax(dim=0)[5]
        loss.backward()
        opt.step()
        losses.append(loss.item())
    print('\nlosses', ['%.4f' % x for x in losses])
    assert losses[-1] < losses[0]
    assert torch.equal(frozen_before, m.edge_attentions[0].attention.key.weight.detach())


def test_module_call_dispatches_on_mode():
    """``model(...)`` is the training forward under train() with autograd on (train_explorer.py:156-176) and the inference
    kernels under eval() or torch.no_grad() (eval_gnn.py:109,168) -- the same rule ModelSmoother.forward follows."""
    from gnnmp.synth import synth_graph
    g = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in synth_graph('maze2', 90, 5, seed=3).items()}
    m = gnnmp.EncoderProcessDecoder(2, 2, 32, 2)
    m.load_state_dict(load_weights('weights_maze'))
    kw = dict(goal=g['goal'], loop=2, v=g['v'], obstacles=g['obstacles'], edge_index=g['edge_index'])
    m.train()
    P_train = m(**kw)
    assert P_train.requires_grad and P_train.grad_fn is not None
    with torch.no_grad():
        P_ng = m(**kw)
    assert not P_ng.requires_grad
    m.eval()
    P_eval = m(**kw)
    assert not P_eval.requires_grad and torch.equal(P_eval, P_ng)
    assert torch.allclose(P_train.detach(), P_eval, rtol=1e-5, atol=2e-5)
    # reduced-precision operands have no training path: under train() with autograd on, the call says so instead of
    # returning an inference result whose backward() fails far from the cause
    m.train()
    m.mlp_dtype = 'bf16'
    with pytest.raises(RuntimeError, match='training path runs in fp32 only'):
        m(**kw)
    with torch.no_grad():
        assert not m(**kw).requires_grad                 # inference under no_grad is fine in any mode


def test_gradients_are_deterministic_and_batched_equals_accumulated():
    """No float atomics in the training kernels: two backward passes on the same inputs give bit-identical gradients; and the
    reference's gradient accumulation over problems (train_explorer.py:184: eight single-graph backward passes per optimizer
    step) equals ONE batched forward / backward of the same problems up to summation order."""
    from gnnmp.synth import synth_graph
    graphs = [{k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in synth_graph('maze2', 150 + 30 * i, 5, seed=20 + i).items()} for i in range(4)]
    m = gnnmp.EncoderProcessDecoder(2, 2, 32, 2)
    m.load_state_dict(load_weights('weights_maze'))
    m.train()
    params = [(n, p) for n, p in m.named_parameters() if n.split('.')[0] in TRAINABLE]
    batch = gnnmp.GraphBatch.from_graphs(graphs, 2, DEV)
    coef = torch.linspace(-1, 1, batch.total_edges, device=DEV)
    runs = []
    for _ in range(2):
        m.zero_grad()
        (m.train_scores(batch, 3) * coef).sum().backward()
        runs.append({n: p.grad.clone() for n, p in params if p.grad is not None})
    assert len(runs[0]) >= 20
    params = [(n, p) for n, p in params if n in runs[0]]
    for n, _ in params:
        assert torch.equal(runs[0][n], runs[1][n]), n
    m.zero_grad()
    off = 0
    for g in graphs:                                                     # one problem per backward, gradients accumulate
        b1 = m._single(g['goal'], g['v'], g['obstacles'], g['edge_index'])
        e = g['edge_index'].shape[1]
        (m.train_scores(b1, 3) * coef[off:off + e]).sum().backward()
        off += e
    for n, p in params:
        scale = float(runs[0][n].abs().max()) + 1e-30
        assert float((p.grad - runs[0][n]).abs().max()) <= 2e-4 * scale + 1e-6, n      # fp32 sums over thousands of rows in another order

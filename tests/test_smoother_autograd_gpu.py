"""Training path of the smoother: the reference trains ModelSmoother under model.train()
(train_smoother.py:33-61), i.e. BatchNorm with batch statistics inside every loop iteration and gradients through the
loop's path updates.  The HIP forward/backward is compared with torch.autograd through the CPU oracle in training mode,
fp64-anchored like the explorer's:

    forward:   allclose(rtol 1e-5, atol 1e-5) against the oracle in fp32 and fp64
    gradients: |g_gpu - g_oracle64| <= max(1e-4 * max|g_oracle64|, 4 * own) + 1e-6 per parameter tensor,
               own = max|g_oracle32 - g_oracle64|
    BatchNorm running statistics after the call = what nn.BatchNorm1d leaves (momentum 0.1, unbiased variance, one update
    per loop iteration)."""
import os

import numpy as np
import pytest
import torch

from conftest import golden_files, load_weights
import gnnmp
from gnnmp.smoother import SMOOTHER_TRAINABLE
from oracle import ref_cpu
from test_smoother_parity import CONF, chain_edges, make

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
FILES = [p for p in golden_files('smoother_') if 'knn32' not in p]


def _oracle(w, r, scale, loss_fn, dtype, loop):
    wd = {k: (t.to(dtype).clone().requires_grad_(True) if t.is_floating_point() and 'running' not in k else t)
          for k, t in w.items()}
    run = (w['node_code.1.running_mean'].to(dtype).clone(), w['node_code.1.running_var'].to(dtype).clone())
    taps = {'bn_running': run}
    out = ref_cpu.smoother_forward(wd, torch.from_numpy(r['path']).to(dtype), torch.from_numpy(r['free']).to(dtype),
                                   torch.from_numpy(r['collided']).to(dtype), torch.from_numpy(r['edge_index']), loop,
                                   scale, taps=taps, training=True)
    loss_fn(out).backward()
    return out.detach(), {k: t.grad for k, t in wd.items() if torch.is_tensor(t) and t.requires_grad}, run


@pytest.mark.parametrize('path', FILES, ids=os.path.basename)
@pytest.mark.parametrize('loop_override', [None, 3])
def test_gradients_match_oracle(path, loop_override):
    with np.load(path) as f:
        r = {k: f[k] for k in f.files}
    name = os.path.basename(path).split('_P')[0].replace('smoother_', '')
    loop = int(r['loop']) if loop_override is None else loop_override
    if loop_override is not None and loop == int(r['loop']):
        pytest.skip('same as the recorded loop count')
    C, scale = CONF[name]
    w = load_weights(name)
    m = make(name)
    m.train()
    P = r['path'].shape[0]
    target = torch.from_numpy(r['out_fp64'])                                 # any fixed target: the loss form of :59

    def loss(o):
        return torch.nn.functional.mse_loss(target.to(o.dtype).to(o.device)[1:-1], o[1:-1])

    out = m(path=torch.from_numpy(r['path']).to(DEV), free=torch.from_numpy(r['free']).to(DEV),
            collided=torch.from_numpy(r['collided']).to(DEV), obstacles=None,
            edge_index=torch.from_numpy(r['edge_index']).to(DEV), loop=loop)
    assert out.requires_grad and out.shape == (P, C)
    loss(out).backward()
    o64, g64, run64 = _oracle(w, r, scale, loss, torch.float64, loop)
    o32, g32, _ = _oracle(w, r, scale, loss, torch.float32, loop)
    e = (out.detach().cpu().double() - o64).abs().max().item()
    assert torch.allclose(out.detach().cpu().double(), o64, rtol=1e-5, atol=1e-5), e
    assert torch.allclose(out.detach().cpu(), o32, rtol=1e-5, atol=1e-5)
    worst = 0.0
    sd = m.state_dict(keep_vars=True)
    trained = {id(sd[k]) for k in SMOOTHER_TRAINABLE}                        # bn2 IS node_code.1 (model_smoother.py:63,65)
    for k, p in sd.items():
        if not isinstance(p, torch.nn.Parameter) or (k not in SMOOTHER_TRAINABLE and id(p) in trained):
            continue
        if k not in SMOOTHER_TRAINABLE:
            assert p.grad is None, k
            continue
        assert p.grad is not None, k
        ref = g64[k]
        own = (g32[k].double() - ref).abs().max().item()
        bar = max(1e-4 * ref.abs().max().item(), 4 * own) + 1e-6
        err = (p.grad.cpu().double() - ref).abs().max().item()
        worst = max(worst, err / bar)
        assert err <= bar, (k, err, bar, own)
    print('\n%s loop %d: forward err %.2e, worst gradient err/bar %.2f' % (os.path.basename(path), loop, e, worst))
    bn = m.node_code[1]
    assert int(bn.num_batches_tracked) == int(w['node_code.1.num_batches_tracked']) + loop
    assert torch.allclose(bn.running_mean.cpu().double(), run64[0], rtol=1e-5, atol=1e-6)
    assert torch.allclose(bn.running_var.cpu().double(), run64[1], rtol=1e-5, atol=1e-6)


def test_eval_mode_and_no_grad_take_the_inference_path():
    path = FILES[0]
    with np.load(path) as f:
        r = {k: f[k] for k in f.files}
    name = os.path.basename(path).split('_P')[0].replace('smoother_', '')
    m = make(name)
    args = dict(path=torch.from_numpy(r['path']).to(DEV), free=torch.from_numpy(r['free']).to(DEV),
                collided=torch.from_numpy(r['collided']).to(DEV), obstacles=None,
                edge_index=torch.from_numpy(r['edge_index']).to(DEV), loop=int(r['loop']))
    m.eval()
    a = m(**args)
    assert not a.requires_grad
    m.train()
    with torch.no_grad():
        b = m(**args)
    assert torch.equal(a, b)                                                 # the reference's eval() numbers either way
    c = m(**args)
    assert c.requires_grad and not torch.equal(a, c)                         # batch statistics change the numbers


def test_sgd_step_moves_the_loss_down():
    """One optimizer step the way train_smoother.py:33-61 takes it (loss over a few problems, one backward)."""
    name = 'smooth_2d_attv3'
    m = make(name)
    m.train()
    sd = m.state_dict(keep_vars=True)
    opt = torch.optim.SGD(list({id(sd[k]): sd[k] for k in SMOOTHER_TRAINABLE}.values()), lr=1e-3)
    gen = torch.Generator().manual_seed(3)
    probs = []
    for P in (8, 12, 20):
        path = torch.rand(P, 2, generator=gen) * 2 - 1
        tgt = path.clone()
        tgt[1:-1] = 0.5 * (path[:-2] + path[2:])
        probs.append(dict(path=path.to(DEV), free=(torch.rand(60, 2, generator=gen) * 2 - 1).to(DEV),
                          collided=(torch.rand(40, 2, generator=gen) * 2 - 1).to(DEV), obstacles=None,
                          edge_index=chain_edges(P).to(DEV), loop=2, target=tgt.to(DEV)))

    def total():
        loss = 0.
        for q in probs:
            a = {k: v for k, v in q.items() if k != 'target'}
            loss = loss + torch.nn.functional.mse_loss(q['target'][1:-1], m(**a)[1:-1])
        return loss / len(probs)

    before = total()
    opt.zero_grad()
    before.backward()
    opt.step()
    m.refresh_weights()
    after = total()
    assert after.item() < before.item(), (before.item(), after.item())


def test_training_steps_keep_the_native_handle_and_momentum_none():
    """BatchNorm's running statistics change after every training forward; the training path never reads them, so the native
    handle (packed + raw weights on the device) must survive across training forwards as long as no trained weight changed,
    and be rebuilt by the next inference call (which folds the statistics).  momentum=None follows torch: cumulative average."""
    gen = torch.Generator().manual_seed(3)
    path, free, coll = (torch.rand(n, 2, generator=gen).to(DEV) * 2 - 1 for n in (10, 50, 40))
    from gnnmp.planner import chain_edge_index
    ei = chain_edge_index(10).to(DEV)
    m = gnnmp.ModelSmoother(workspace_size=2, config_size=2, embed_size=128, obs_size=6)
    m.load_state_dict(load_weights('smooth_2d_attv3'))
    m.train()
    m(path=path, free=free, collided=coll, edge_index=ei, loop=2)
    h1 = m._handle
    m(path=path, free=free, collided=coll, edge_index=ei, loop=2)          # running stats were bumped in between
    assert m._handle is h1
    m.eval()
    with torch.no_grad():
        m(path=path, free=free, collided=coll, edge_index=ei, loop=1)      # inference folds the new statistics: rebuild
    assert m._handle is not h1
    # momentum None: cumulative moving average, like nn.BatchNorm1d
    ref = torch.nn.BatchNorm1d(128, momentum=None).to(DEV).train()
    m2 = gnnmp.ModelSmoother(workspace_size=2, config_size=2, embed_size=128, obs_size=6)
    m2.load_state_dict(load_weights('smooth_2d_attv3'))
    m2.node_code[1].momentum = None
    m2.node_code[1].reset_running_stats()
    m2.train()
    nodes = torch.cat((path, free, coll))
    info = torch.zeros(100, 3, device=DEV); info[:10, 0] = 1; info[10:60, 1] = 1; info[60:, 2] = 1
    with torch.no_grad():
        x0 = torch.nn.functional.linear(torch.cat((nodes, info), -1), m2.node_code[0].weight.to(DEV), m2.node_code[0].bias.to(DEV))
    ref(x0)
    m2(path=path, free=free, collided=coll, edge_index=ei, loop=1)
    assert int(m2.node_code[1].num_batches_tracked) == 1
    assert torch.allclose(m2.node_code[1].running_mean.cpu(), ref.running_mean.cpu(), rtol=1e-5, atol=1e-6)
    assert torch.allclose(m2.node_code[1].running_var.cpu(), ref.running_var.cpu(), rtol=1e-4, atol=1e-6)

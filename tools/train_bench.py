#!/usr/bin/env python
"""Run ON THE GPU BOX: the explorer's training step at the reference's shape (train_explorer.py:156-186: eight planning problems
per optimizer step, one forward/backward each, gradients accumulated, Adam) -- milliseconds per optimizer step
  (a) exactly like the reference: 8 x (model(...) -> cross-entropy over a frontier -> backward), then optimizer.step();
  (b) the same eight problems as ONE batched forward/backward (gnnmp.GraphBatch: the loss is a sum over problems);
  (c) the CPU oracle (oracle/ref_cpu.py through torch.autograd, the reference's formulation) for the same eight problems,
and a determinism check: two backward passes on the same inputs give bit-identical gradients."""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import gnnmp  # noqa: E402
from gnnmp.synth import synth_graph  # noqa: E402
from gnnmp.weights import load_weights  # noqa: E402
from oracle import ref_cpu  # noqa: E402  (timed CPU baseline only)

DEV = 'cuda:0'
N, K1, LOOP, NPROB = 1000, 8, 5, 8
TRAINABLE = {'node_code', 'edge_code', 'goal_encoder', 'encoder', 'process', 'decoder', 'policy'}
graphs = [synth_graph('maze2', N, K1, seed=900 + i) for i in range(NPROB)]
dgraphs = [{k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in g.items()} for g in graphs]
w = load_weights('weights_maze')
m = gnnmp.EncoderProcessDecoder(2, 2, 32, 2)
m.load_state_dict(w)
m.to(DEV)                             # train_explorer.py:105: the reference trains with the model on the device
m.train()
params = [p for n, p in m.named_parameters() if n.split('.')[0] in TRAINABLE]
opt = torch.optim.Adam(params, lr=1e-4)
frontier = torch.arange(0, 40, device=DEV)


def loss_of(P):                       # train_explorer.py:170-176 shape: rows of the frontier nodes, log-softmax, one label
    return -P[frontier].reshape(-1).log_softmax(dim=0)[17]


def step_reference_shape():
    opt.zero_grad()
    for g in dgraphs:
        P = m(goal=g['goal'], loop=LOOP, v=g['v'], obstacles=g['obstacles'], edge_index=g['edge_index'])
        (loss_of(P) / NPROB).backward()
    opt.step()


batch = gnnmp.GraphBatch.from_graphs(dgraphs, 2, DEV)
sel = []
off = 0
for g in dgraphs:                     # the same loss on the sparse scores: edges whose TARGET is a frontier node, in dense row-major order
    ei = g['edge_index']
    mask = ei[1] < 40
    sel.append((off, ei, mask))
    off += ei.shape[1]


def step_batched():
    opt.zero_grad()
    s = m.train_scores(batch, LOOP)
    total = 0
    for (o, ei, mask), g in zip(sel, dgraphs):
        P = s.new_zeros(N, N).index_put((ei[1], ei[0]), s[o:o + ei.shape[1]])
        total = total + loss_of(P) / NPROB
    total.backward()
    opt.step()


def timeit(fn, reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


ms_ref = timeit(step_reference_shape, 5)
ms_bat = timeit(step_batched, 5)

# determinism: same inputs, two backward passes
m.load_state_dict(w)
grads = []
for _ in range(2):
    m.zero_grad()
    s = m.train_scores(batch, LOOP)
    (s * torch.linspace(-1, 1, s.numel(), device=DEV)).sum().backward()
    grads.append(torch.cat([p.grad.reshape(-1).clone() for p in params if p.grad is not None]))
identical = bool(torch.equal(grads[0], grads[1]))

# CPU oracle autograd, same problems (one optimizer step = 8 forward/backward)
wcpu = {k: (t.clone().requires_grad_(k.split('.')[0] in TRAINABLE) if t.is_floating_point() else t) for k, t in w.items()}
t0 = time.perf_counter()
for g in graphs[:2]:
    P = ref_cpu.explorer_forward(wcpu, g['v'], g['goal'], g['obstacles'], g['edge_index'], LOOP, dense=True, detach=True)
    (-P[:40].reshape(-1).log_softmax(dim=0)[17] / NPROB).backward()
cpu_ms = (time.perf_counter() - t0) / 2 * NPROB * 1e3

print('explorer training step, %d problems of %d nodes (k1 = %d, E ~ %d), loop %d, weights_maze, fp32, one MI355X'
      % (NPROB, N, K1, graphs[0]['edge_index'].shape[1], LOOP))
print('  (a) reference shape: 8 x (module call, loss, backward) + Adam step   %8.2f ms / optimizer step' % ms_ref)
print('  (b) one batched forward / backward of the 8 problems + Adam step      %8.2f ms / optimizer step' % ms_bat)
print('  (c) CPU oracle through torch.autograd (%d torch threads), extrapolated from 2 problems  %8.0f ms / optimizer step'
      % (torch.get_num_threads(), cpu_ms))
print('  gradients of two identical backward passes bit-identical: %s' % identical)

# ---- the smoother's training step in the reference's shape (train_smoother.py:33-61): eight replay entries per optimizer step, one
# forward each under model.train() (BatchNorm with batch statistics), loop drawn from 1..9 (fixed to 5 here), MSE on the inner
# waypoints, ONE backward of the summed loss, SGD with momentum (train_smoother.py:81)
from gnnmp.planner import chain_edge_index  # noqa: E402
ws_ = load_weights('smooth_2d_attv3')
sm = gnnmp.ModelSmoother(workspace_size=2, config_size=2, embed_size=128, obs_size=6)
sm.load_state_dict(ws_)
sm.to(DEV)
sm.train()
sopt = torch.optim.SGD(sm.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
gen = torch.Generator().manual_seed(11)
replay = []
for _ in range(NPROB):
    P_ = 20
    path = (torch.rand(P_, 2, generator=gen) * 2 - 1).to(DEV)
    free, coll = (torch.rand(500, 2, generator=gen) * 2 - 1).to(DEV), (torch.rand(500, 2, generator=gen) * 2 - 1).to(DEV)
    target = (path + 0.05 * torch.randn(P_, 2, generator=gen).to(DEV))
    replay.append((path, free, coll, chain_edge_index(P_).to(DEV), target))


def smoother_step():
    sopt.zero_grad()
    loss = 0.
    for path, free, coll, ei, target in replay:
        pred = sm(path=path, free=free, collided=coll, obstacles=None, edge_index=ei, loop=5)
        loss = loss + torch.nn.MSELoss()(target[1:-1], pred[1:-1])
    (loss / len(replay)).backward()
    sopt.step()


ms_sm = timeit(smoother_step, 5)
print('smoother training step (train_smoother.py:33-61 shape: 8 replay entries of 20 waypoints + 500 + 500 samples, loop 5, smooth_2d_attv3,\n'
      '  model.train(), one backward of the summed MSE, SGD + momentum)       %8.2f ms / optimizer step' % ms_sm)

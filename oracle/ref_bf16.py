"""CPU emulation of the kernels' bf16 mode (TEST INFRASTRUCTURE ONLY, like everything under oracle/).

``gnnmp_explorer_dims.mlp_dtype = GNNMP_BF16`` rounds every MFMA operand (weights, activations,
obstacle keys / values, softmax weights) to bf16 (round-to-nearest-even) and accumulates in fp32;
bias, ReLU, LayerNorm, softmax statistics, residuals and the max aggregation stay fp32.  Unlike
``ref_cpu`` this file follows the KERNELS' formulation (SURVEY.md Appendix E rewrites: first layers over
concatenations split per operand, with the operand-combined matrices such as W_a + W_b rounded after the
fp32 sum), because in bf16 the rounding points are part of the contract.  It answers "does the kernel
do what the bf16 mode claims"; accuracy against the fp32 reference is judged separately with the
statistical bar of SURVEY.md section 7.3.1.
"""
import math

import torch
import torch.nn.functional as F

from .ref_cpu import knn, scatter_rows


def r(x):
    return x.bfloat16().float()


def lin(x, W, b=None):
    y = r(x) @ r(W).T
    return y if b is None else y + b


def _mlp2(w, name, x):
    return lin(F.relu(lin(x, w[name + '.0.weight'], w[name + '.0.bias'])), w[name + '.2.weight'], w[name + '.2.bias'])


def _ln(w, name, x):
    return F.layer_norm(x, (x.shape[-1],), w[name + '.weight'], w[name + '.bias'], 1e-6)


def _ffn(w, pre, x):
    y = lin(F.relu(lin(x, w[pre + '.w_1.weight'], w[pre + '.w_1.bias'])), w[pre + '.w_2.weight'], w[pre + '.w_2.bias'])
    return _ln(w, pre + '.layer_norm', y + x)


def _wqk(w, a):
    """The kernels' combined query-key matrix: Wq^T Wk, product in double, rounded to fp32 once (api.cpp)."""
    return (w[a + '.query.weight'].double().T @ w[a + '.key.weight'].double()).float()


def _attention(w, pre, m, ko, vo):
    """ko, vo: bf16-rounded obstacle keys / values [O, d] (as stored in the K/V slabs).  Online softmax over
    32-obstacle tiles exactly like attention_block (p is rounded relative to the running maximum)."""
    d = m.shape[1]
    a = pre + '.attention'
    tq = lin(m, _wqk(w, a))                                  # Wqk m with Wqk = Wq^T Wk (layout.hpp AttBlob)
    acc = lin(m, w[a + '.value.weight'])
    mx = (m * tq).sum(-1)                                    # self logit m . (Wqk m)
    psum = torch.ones_like(mx)
    cs = math.log2(math.e) / math.sqrt(d)
    qb = r(m)                                                # obstacle keys are premultiplied: ko = r(Wqk code)
    for o0 in range(0, ko.shape[0], 32):
        s = qb @ ko[o0:o0 + 32].T
        nmx = torch.maximum(mx, s.max(-1).values)
        alpha = torch.exp2((mx - nmx) * cs)
        p = torch.exp2((s - nmx.unsqueeze(-1)) * cs)
        psum = psum * alpha + p.sum(-1)
        acc = acc * alpha.unsqueeze(-1) + r(p) @ vo[o0:o0 + 32]
        mx = nmx
    new = acc / psum.unsqueeze(-1) + m
    m = _ln(w, a + '.layer_norm', new)
    return _ffn(w, pre + '.map_feed', m)


@torch.no_grad()
def explorer_forward_bf16(w, v, goal, obstacles, edge_index, loop, use_obstacles=True):
    C = v.shape[1]
    d = w['encoder.bias'].shape[0]
    n = v.shape[0]
    s, t = edge_index[0], edge_index[1]
    g = goal.view(-1, C)
    nc = _mlp2(w, 'node_code', torch.cat((v, g.repeat(n, 1), (v - g) ** 2, v - g), dim=-1))
    pair = torch.cat((v[s], v[t]), dim=-1)
    # W1e . edge_code + b1 with the product folded into the encoder's second layer in double (api.cpp pack_explorer)
    w1e = w['process.lin_0.0.weight'][:, 4 * d:].double()
    fold = (w1e @ w['edge_code.2.weight'].double()).float()
    fold_b = (w1e @ w['edge_code.2.bias'].double() + w['process.lin_0.0.bias'].double()).float()
    ec = lin(F.relu(lin(pair, w['edge_code.0.weight'], w['edge_code.0.bias'])), fold, fold_b)
    nf = _mlp2(w, 'node_free_code', v)
    ef = _mlp2(w, 'edge_free_code', pair)
    if use_obstacles:
        S = w['obs_node_code.0.weight'].shape[1]
        ob = obstacles.reshape(-1, S)
        on, oe = _mlp2(w, 'obs_node_code', ob), _mlp2(w, 'obs_edge_code', ob)
        for b in range(3):
            pn, pe = 'node_attentions.%d' % b, 'edge_attentions.%d' % b
            nf = _attention(w, pn, nf, r(lin(on, _wqk(w, pn + '.attention'))), r(lin(on, w[pn + '.attention.value.weight'])))
            ef = _attention(w, pe, ef, r(lin(oe, _wqk(w, pe + '.attention'))), r(lin(oe, w[pe + '.attention.value.weight'])))
            on, oe = _ffn(w, pn + '.obs_feed', on), _ffn(w, pe + '.obs_feed', oe)
    gi = int(knn(v, g, 1)[1][0])
    we, wd, w1, p0, wl1 = (w['encoder.weight'], w['decoder.weight'], w['process.lin_0.0.weight'], w['policy.0.weight'],
                           w['process.lin_1.weight'])
    ge = w['goal_encoder']
    weg, wehg = we[:, 2 * d:3 * d] @ ge, we[:, 3 * d:] @ ge                  # host fp32 matvecs (api.cpp)
    wsrc = w1[:, :d] + w1[:, d:2 * d]
    wdst = w1[:, 2 * d:3 * d] - w1[:, :d]
    wps, wpt = p0[:, :d] + p0[:, d:2 * d], p0[:, d:2 * d]
    xi = lin(nc, we[:, :d]) + lin(nf, we[:, d:2 * d]) + w['encoder.bias']
    xi[gi] = xi[gi] + weg
    x = xi.clone()
    x[gi] = x[gi] + wehg
    # A, B, K_e, PE (and the policy's node terms) are STORED in bf16 in this mode (chain.hpp store_*_p)
    A, B = r(lin(x, wsrc)), r(lin(x, wdst))
    dn = lin(nc, wd[:, :d]) + w['decoder.bias']
    ke = r(lin(ef, w1[:, 3 * d:4 * d]) + ec)
    pe_ = r(lin(ef, p0[:, 2 * d:]) + w['policy.0.bias'])
    for it in range(loop):
        hid = F.relu(A[s] + B[t] + ke)
        msg = lin(hid, w['process.lin_0.2.weight'], w['process.lin_0.2.bias'])
        agg = scatter_rows(msg, t, n, 'max')
        h = lin(x, wl1[:, :d]) + lin(agg, wl1[:, d:]) + w['process.lin_1.bias']
        if it < loop - 1:
            x = xi + lin(h, we[:, 3 * d:])
            A, B = r(lin(x, wsrc)), r(lin(x, wdst))
        else:
            dec = dn + lin(h, wd[:, d:])
            A, B = r(lin(dec, wps)), r(lin(dec, wpt))
    h1 = F.relu(A[s] - B[t] + pe_)
    h2 = F.relu(lin(h1, w['policy.2.weight'], w['policy.2.bias']))
    return (h2 * w['policy.4.weight'].view(1, -1)).sum(-1)


@torch.no_grad()
def smoother_forward_bf16(w, path, free, collided, edge_index, loop=1, scale=1.0):
    """bf16-operand emulation of the smoother kernels' formulation (BatchNorm folded into node_code.0 in fp32,
    lin_0 split into (W_a + W_b) x_j + (W_c - W_a) x_i, ordered fp32 sum of the messages)."""
    from .ref_cpu import coalesce
    path = path / scale
    free = free / scale
    collided = collided / scale
    P = path.shape[0]
    d = w['node_code.3.bias'].shape[0]
    nodes = torch.cat((path, free, collided), dim=0)
    n = nodes.shape[0]
    g = w['node_code.1.weight'] / torch.sqrt(w['node_code.1.running_var'] + 1e-5)
    w0 = w['node_code.0.weight'] * g.unsqueeze(1)
    b0 = (w['node_code.0.bias'] - w['node_code.1.running_mean']) * g + w['node_code.1.bias']
    w1 = w['process.lin_0.0.weight']
    wsrc, wdst = w1[:, :d] + w1[:, d:2 * d], w1[:, 2 * d:] - w1[:, :d]
    for _ in range(loop):
        ne = knn(nodes[P:], path, 10).flip(0)
        ne[0, :] = ne[0, :] + P
        ei = coalesce(torch.cat((edge_index, ne), dim=-1), n)
        ei = ei[:, ei[1] < P]
        info = nodes.new_zeros(n, 3)
        info[:P, 0] = 1
        info[P:P + free.shape[0], 1] = 1
        info[P + free.shape[0]:, 2] = 1
        x = lin(F.relu(lin(torch.cat((nodes, info), dim=-1), w0, b0)), w['node_code.3.weight'], w['node_code.3.bias'])
        s, t = ei[0], ei[1]
        z = F.relu(lin(x[t], wdst) + lin(x[s], wsrc) + w['process.lin_0.0.bias'])
        msg = lin(z, w['process.lin_0.2.weight'], w['process.lin_0.2.bias'])
        S = x.new_zeros(P, d)
        for e in range(ei.shape[1]):                      # coalesced order: fp32 sequential sum per target
            S[t[e]] = S[t[e]] + msg[e]
        hh = x[:P] + lin(F.relu(lin(S, w['process.lin_1.0.weight'], w['process.lin_1.0.bias'])),
                         w['process.lin_1.2.weight'], w['process.lin_1.2.bias'])
        new = lin(hh, w['smooth_node.weight'], w['smooth_node.bias'])
        path = path.clone()
        path[1:-1] = new[1:-1]
        nodes = nodes.clone()
        nodes[:P] = path
    return path * scale

"""CPU oracle for the GNN explorer / smoother forward passes.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it,
and there only as the checker / the timed CPU baseline.  The product path
(``gnn-motion-planning_amd``) never routes through this file and raises if the HIP library
is missing.

What it is: a plain-PyTorch (CPU) restatement, in the *reference's own formulation*
(materialised gathers and concatenations, same operation order, no algebraic rewrites), of

  * ``EncoderProcessDecoder.forward``   reference ``model.py:115-150``
      (``MPNN`` ``model.py:22-41``, ``Attention`` ``:153-181``, ``FeedForward`` ``:184-201``,
       ``Block`` ``:204-218``)
  * ``ModelSmoother.forward``           reference ``model_smoother.py:104-142``
      (``MPNN`` add-aggregation ``model_smoother.py:22-39``)

and of the third-party primitives those call whose source is not in the reference tree
(``torch_geometric`` ``MessagePassing.propagate`` / ``knn`` / ``knn_graph``,
``torch_scatter.scatter``, ``torch_sparse.coalesce``; versions unpinned by the reference,
see SURVEY.md section 8(c)); their semantics are restated from their published behaviour.

Pinning: the restatement is checked (``tests/test_oracle_golden.py``) against golden
vectors produced in the authoring container by importing the *unmodified* reference modules
(``tools/gen_golden.py``) for every shipped explorer / smoother checkpoint, in fp32 and fp64.
At the PyG boundary itself the reference holds no tests or golden vectors, so that boundary
is pinned only by (i) those generated goldens and (ii) the published notebook known-answer
(``main.ipynb:57-61``) which the same stand-in semantics reproduce (SURVEY.md finding 0.5).

Everything takes a ``state_dict``-like mapping ``w`` (name -> tensor) with the reference's
parameter names, so the shipped ``.pt`` files are used unchanged.
"""
import math

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# third-party primitives, restated
# --------------------------------------------------------------------------------------
def scatter_rows(msg, index, n_rows, reduce):
    """torch_scatter.scatter(msg, index, dim=0, dim_size=n_rows, reduce=...).

    Rows never written stay 0 (torch_scatter fills untouched slots with 0, not -inf);
    behind ``MessagePassing.propagate`` at model.py:33 ('max') and model_smoother.py:32 ('add').
    """
    out = msg.new_zeros((n_rows, msg.shape[1]))
    if index.numel() == 0:
        return out
    idx = index.view(-1, 1).expand_as(msg)
    red = {'max': 'amax', 'add': 'sum'}[reduce]
    return out.scatter_reduce(0, idx, msg, reduce=red, include_self=False)


def knn(x, y, k, input_dtype=False):
    """torch_geometric.nn.pool.knn(x, y, k): for each row of y the k nearest rows of x.

    Returns int64 [2, len(y)*min(k, len(x))]; row 0 indexes y, row 1 indexes x
    (call sites model.py:132, model_smoother.py:125).  Distances in float64,
    ``topk(largest=False)`` ordering.  ``input_dtype``: squared distances accumulated coordinate by coordinate in
    the dtype of the inputs instead (a float32 kNN library's arithmetic; torch_cluster's dtype is unpinned by the
    reference), ties to the lower index -- pinned by tests/golden/smoother_*_knn32.npz.
    """
    k = min(k, x.shape[0])
    if input_dtype:
        d = torch.zeros(y.shape[0], x.shape[0], dtype=x.dtype)
        for c in range(x.shape[1]):
            df = x[:, c].view(1, -1) - y[:, c].view(-1, 1)
            d = d + df * df
        nb = torch.sort(d, dim=1, stable=True).indices[:, :k]
        q = torch.arange(y.shape[0]).view(-1, 1).expand_as(nb)
        return torch.stack((q.reshape(-1), nb.reshape(-1)), dim=0)
    d = torch.cdist(y.to(torch.float64), x.to(torch.float64))
    nb = d.topk(k, dim=1, largest=False).indices
    q = torch.arange(y.shape[0]).view(-1, 1).expand_as(nb)
    return torch.stack((q.reshape(-1), nb.reshape(-1)), dim=0)


def knn_graph(x, k, loop=True):
    """torch_geometric.nn.knn_graph: row 0 = neighbour (source), row 1 = centre (target)
    (call sites eval_gnn.py:160,162)."""
    e = knn(x, x, k if loop else k + 1)
    src, dst = e[1], e[0]
    if not loop:
        keep = src != dst
        src, dst = src[keep], dst[keep]
    return torch.stack((src, dst), dim=0)


def coalesce(edge_index, n):
    """torch_sparse.coalesce(index, None, n, n): sort columns by (row0, row1), drop duplicates
    (call sites eval_gnn.py:164, model_smoother.py:128)."""
    key = edge_index[0].to(torch.int64) * n + edge_index[1].to(torch.int64)
    key = torch.unique(key, sorted=True)
    return torch.stack((key // n, key % n), dim=0)


# --------------------------------------------------------------------------------------
# small building blocks
# --------------------------------------------------------------------------------------
def _lin(w, name, x, bias=True):
    return F.linear(x, w[name + '.weight'], w[name + '.bias'] if bias else None)


def _mlp2(w, name, x):
    """Seq(Lin, ReLU, Lin): modules ``<name>.0`` and ``<name>.2``."""
    return _lin(w, name + '.2', F.relu(_lin(w, name + '.0', x)))


def _layer_norm(w, name, x, eps):
    return F.layer_norm(x, (x.shape[-1],), w[name + '.weight'], w[name + '.bias'], eps)


def _attention(w, pre, m, o, materialize=False):
    """``Attention.forward`` (model.py:164-181); temperature sqrt(d) (model.py:208).

    ``materialize=True`` forms the [rows, O+1, d] product tensor exactly as model.py:178-179
    does (that materialisation is ~80 % of the reference's CPU time, SURVEY.md finding 0.8);
    the default contracts it as a matmul (same sum, different fp32 association).  The
    bench's ``cpu_baseline`` times the materialising form because that is what the
    reference's CPU path costs."""
    d = m.shape[1]
    mv = _lin(w, pre + '.value', m, bias=False)
    ov = _lin(w, pre + '.value', o, bias=False)
    mq = _lin(w, pre + '.query', m, bias=False)
    mk = _lin(w, pre + '.key', m, bias=False)
    ok = _lin(w, pre + '.key', o, bias=False)
    obs_att = mq @ ok.T                                         # model.py:173
    self_att = (mq * mk).sum(dim=-1)                            # model.py:174
    att = torch.cat((self_att.unsqueeze(-1), obs_att), dim=-1)  # model.py:175
    att = (att / (d ** 0.5)).softmax(dim=-1)                    # model.py:176
    if materialize:                                             # model.py:178-179
        vals = torch.cat((mv.unsqueeze(1), ov.unsqueeze(0).repeat(len(m), 1, 1)), dim=1)
        new = (att.unsqueeze(-1) * vals).sum(dim=1)
    else:
        new = att[:, :1] * mv + att[:, 1:] @ ov                 # same sum, contracted
    return _layer_norm(w, pre + '.layer_norm', new + m, 1e-6)   # model.py:181


def _feed_forward(w, pre, x):
    """``FeedForward.forward`` (model.py:192-201)."""
    y = _lin(w, pre + '.w_2', F.relu(_lin(w, pre + '.w_1', x)))
    return _layer_norm(w, pre + '.layer_norm', y + x, 1e-6)


def _block(w, pre, m, o, materialize=False):
    """``Block.forward`` (model.py:212-218): obstacle rows only see their own FFN."""
    m = _attention(w, pre + '.attention', m, o, materialize)
    m = _feed_forward(w, pre + '.map_feed', m)
    o = _feed_forward(w, pre + '.obs_feed', o)
    return m, o


# --------------------------------------------------------------------------------------
# explorer
# --------------------------------------------------------------------------------------
def explorer_forward(w, v, goal, obstacles, edge_index, loop, use_obstacles=True,
                     obs_size=None, taps=None, dense=False, materialize=False, detach=False):
    """Inference restatement (eval_gnn.py:168 runs it under no_grad); with ``detach`` the autograd graph is kept and cut
    where the reference cuts it -- see :func:`_explorer_forward`."""
    if detach:
        return _explorer_forward(w, v, goal, obstacles, edge_index, loop, use_obstacles, obs_size, taps, dense, materialize, True)
    with torch.no_grad():
        return _explorer_forward(w, v, goal, obstacles, edge_index, loop, use_obstacles, obs_size, taps, dense, materialize, False)


def _explorer_forward(w, v, goal, obstacles, edge_index, loop, use_obstacles=True,
                      obs_size=None, taps=None, dense=False, materialize=False, detach=False):
    """``EncoderProcessDecoder.forward`` (model.py:115-150).

    Returns per-edge scores [E] in the order of ``edge_index`` columns, or the dense
    ``P[target, source]`` matrix (model.py:148-149) when ``dense``.
    ``taps``: optional dict that receives intermediates for kernel-level diffing.
    ``detach``: cut the autograd graph where the reference does (model.py:141,142,146: node_free_code and
    edge_free_code are detached before every use) -- the gradient oracle of the training path.
    """
    C = v.shape[1]
    d = w['encoder.bias'].shape[0]
    n = v.shape[0]
    s, t = edge_index[0], edge_index[1]
    g = goal.view(-1, C)                                                      # :117
    node_code = _mlp2(w, 'node_code',
                      torch.cat((v, g.repeat(n, 1), (v - g) ** 2, v - g), dim=-1))  # :119
    pair = torch.cat((v[s], v[t]), dim=-1)
    edge_code = _mlp2(w, 'edge_code', pair)                                   # :120
    nf = _mlp2(w, 'node_free_code', v)                                        # :122
    ef = _mlp2(w, 'edge_free_code', pair)                                     # :123
    if use_obstacles:                                                         # :125
        S = obs_size if obs_size is not None else w['obs_node_code.0.weight'].shape[1]
        ob = obstacles.reshape(-1, S)
        on = _mlp2(w, 'obs_node_code', ob)                                    # :126
        oe = _mlp2(w, 'obs_edge_code', ob)                                    # :127
        for b in range(3):                                                    # :128-130
            nf, on = _block(w, 'node_attentions.%d' % b, nf, on, materialize)
            ef, oe = _block(w, 'edge_attentions.%d' % b, ef, oe, materialize)
    if detach:
        nf, ef = nf.detach(), ef.detach()
    gi = knn(v, g, 1)[1]                                                      # :132
    h0 = node_code.new_zeros(n, d)                                            # :133
    h0[gi, :] = h0[gi, :] + w['goal_encoder']                                 # :134
    h = h0
    if loop < 1:
        raise ValueError('loop must be >= 1 (decode is only bound inside the loop, model.py:139-145)')
    hs = []
    for _ in range(loop):                                                     # :139
        x = _lin(w, 'encoder', torch.cat((node_code, nf, h0, h), dim=-1))     # :141
        xj, xi = x[s], x[t]
        z = torch.cat((xj - xi, xj, xi, ef, edge_code), dim=-1)               # :38-39, :142
        msg = _mlp2(w, 'process.lin_0', z)                                    # :40
        agg = scatter_rows(msg, t, n, 'max')                                  # :33
        h = _lin(w, 'process.lin_1', torch.cat((x, agg), dim=-1))             # :36
        dec = _lin(w, 'decoder', torch.cat((node_code, h), dim=-1))           # :143
        hs.append(h)
    pin = torch.cat((dec[s], dec[s] - dec[t], ef), dim=-1)                    # :145
    p = F.relu(_lin(w, 'policy.0', pin))
    p = F.relu(_lin(w, 'policy.2', p))
    score = F.linear(p, w['policy.4.weight']).squeeze(-1)                     # :146 (no bias)
    if taps is not None:
        taps.update(node_code=node_code, edge_code=edge_code, node_free_code=nf,
                    edge_free_code=ef, goal_index=gi, h=hs, decode=dec)
    if dense:
        out = score.new_zeros(n, n)                                           # :148
        out[t, s] = score                                                     # :149
        return out
    return score


# --------------------------------------------------------------------------------------
# smoother
# --------------------------------------------------------------------------------------
def smoother_forward(w, path, free, collided, edge_index, loop=1, scale=1.0, taps=None, knn_input_dtype=False, training=False):
    """Inference restatement (smoother.py:233 runs it under eval()); ``training`` = the reference's training call
    (train_smoother.py:52 under model.train()): BatchNorm with batch statistics, autograd graph kept."""
    if training:
        return _smoother_forward(w, path, free, collided, edge_index, loop, scale, taps, knn_input_dtype, True)
    with torch.no_grad():
        return _smoother_forward(w, path, free, collided, edge_index, loop, scale, taps, knn_input_dtype, False)


def _smoother_forward(w, path, free, collided, edge_index, loop=1, scale=1.0, taps=None, knn_input_dtype=False, training=False):
    """``ModelSmoother.forward`` (model_smoother.py:104-142); ``obstacles`` is accepted and
    ignored by the reference, so it is not a parameter here."""
    path = path / scale                                                       # :118
    free = free / scale
    collided = collided / scale
    P = path.shape[0]
    nodes = torch.cat((path, free, collided), dim=0)                          # :121
    n = nodes.shape[0]
    for _ in range(loop):                                                     # :123
        ne = knn(nodes[P:], path, 10, knn_input_dtype).flip(0)                # :125
        ne[0, :] = ne[0, :] + P                                               # :126
        ei = coalesce(torch.cat((edge_index, ne), dim=-1), n)                 # :127-128
        info = nodes.new_zeros(n, 3)                                          # :130-133
        info[:P, 0] = 1
        info[P:P + free.shape[0], 1] = 1
        info[P + free.shape[0]:, 2] = 1
        x = _lin(w, 'node_code.0', torch.cat((nodes, info), dim=-1))          # :135-136
        if training:
            run = (taps or {}).get('bn_running')                                  # (mean, var) updated in place like nn.BatchNorm1d
            x = F.batch_norm(x, run[0] if run else None, run[1] if run else None, w['node_code.1.weight'],
                             w['node_code.1.bias'], True, 0.1, 1e-5)
        else:
            x = F.batch_norm(x, w['node_code.1.running_mean'], w['node_code.1.running_var'],
                             w['node_code.1.weight'], w['node_code.1.bias'], False, 0.0, 1e-5)
        x = _lin(w, 'node_code.3', F.relu(x))
        s, t = ei[0], ei[1]
        xj, xi = x[s], x[t]
        msg = _mlp2(w, 'process.lin_0', torch.cat((xj - xi, xj, xi), dim=-1))  # :36-39
        agg = scatter_rows(msg, t, n, 'add')                                  # :32
        hh = x + _mlp2(w, 'process.lin_1', agg)                               # :34
        new = _lin(w, 'smooth_node', hh[:P])
        path = path.clone()
        path[1:-1] = new[1:-1]                                                # :139
        nodes = nodes.clone()
        nodes[:P] = path                                                      # :140
        if taps is not None:
            taps.setdefault('edge_index', []).append(ei)
            taps.setdefault('x', []).append(x)
    return path * scale                                                       # :142


# --------------------------------------------------------------------------------------
# host-side callers restated -- used to build test inputs
# --------------------------------------------------------------------------------------
def build_edges(v, n_free, k1):
    """Edge construction of ``create_data`` (eval_gnn.py:159-164) with k1 given directly:
    kNN(all) + reverse + kNN(free only) + reverse, self loops included, coalesced."""
    e = knn_graph(v, k1, loop=True)
    e = torch.cat((e, e.flip(0)), dim=-1)
    ef = knn_graph(v[:n_free], k1, loop=True)
    e = torch.cat((e, ef, ef.flip(0)), dim=-1)
    return coalesce(e, v.shape[0])


def k1_of(k, n_free):
    """``k1 = int(ceil(k * ln(n_free) / ln(100)))`` (eval_gnn.py:159)."""
    return int(math.ceil(k * math.log(n_free) / math.log(100)))

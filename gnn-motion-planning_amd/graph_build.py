"""Host-side graph construction: the build's counterpart of the reference's ``create_data``
(eval_gnn.py:150-165) and of the PyG primitives it calls (``knn_graph`` eval_gnn.py:160,162;
``coalesce`` eval_gnn.py:164).  Plain CPU tensor code; the planner control flow stays on the
host (north_star), this is SURVEY.md section 8(a) row H1.

Semantics:
  * ``knn_graph(x, k, loop=True)``: for every point the k nearest points *including itself*;
    edge (source = neighbour, target = centre).
  * edges = kNN(all nodes) + reversed + kNN(free nodes only) + reversed, then coalesced:
    columns sorted by (source, target), duplicates dropped.
  * ``k1 = ceil(k * ln(n_free) / ln(100))``.
"""
import math

import torch


def knn_indices(x, k):
    """Indices [n, min(k, n)] of the k nearest rows of ``x`` to each row of ``x`` (self
    included, Euclidean, float64 distances, ascending)."""
    k = min(int(k), x.shape[0])
    xd = x.to(torch.float64)
    return torch.cdist(xd, xd).topk(k, dim=1, largest=False).indices


def knn_graph(x, k):
    """[2, n*k] int64: row 0 = neighbour (message source), row 1 = centre (message target)."""
    nb = knn_indices(x, k)
    centre = torch.arange(x.shape[0]).view(-1, 1).expand_as(nb)
    return torch.stack((nb.reshape(-1), centre.reshape(-1)), dim=0)


def coalesce(edge_index, n):
    """Sort columns lexicographically by (row 0, row 1) and drop duplicate columns."""
    key = torch.unique(edge_index[0].to(torch.int64) * n + edge_index[1].to(torch.int64), sorted=True)
    return torch.stack((key // n, key % n), dim=0)


def k1_of(k, n_free):
    """Neighbour count actually used: eval_gnn.py:159."""
    return int(math.ceil(k * math.log(n_free) / math.log(100)))


def build_edges(v, n_free, k1):
    """Coalesced edge_index of ``create_data`` (eval_gnn.py:160-164) for node matrix ``v``
    whose first ``n_free`` rows are the collision-free samples."""
    e = knn_graph(v, k1)
    ef = knn_graph(v[:n_free], k1)
    return coalesce(torch.cat((e, e.flip(0), ef, ef.flip(0)), dim=1), v.shape[0])


def create_data(free, collided, goal_state, k):
    """Counterpart of ``create_data`` (eval_gnn.py:150-165).  ``free``/``collided`` are
    sequences of configurations (float64 as sampled); returns a dict with the tensors the
    explorer's ``forward`` receives: goal [C], v [N,C] fp32, labels [N,3], edge_index [2,E]."""
    import numpy as np
    vf = torch.tensor(np.asarray(free), dtype=torch.float32).reshape(len(free), -1)
    C = vf.shape[1]
    vc = torch.tensor(np.asarray(collided), dtype=torch.float32).reshape(-1, C)
    v = torch.cat((vf, vc), dim=0)
    labels = torch.zeros(v.shape[0], 3)
    labels[:len(free), 0] = 1
    labels[len(free):, 1] = 1
    labels[1, 2] = 1
    k1 = k1_of(k, len(free))
    return {'goal': torch.tensor(np.asarray(goal_state), dtype=torch.float32), 'v': v,
            'labels': labels, 'edge_index': build_edges(v, len(free), k1)}


# --------------------------------------------------------------------------------------------------
# device-side construction (libgnnmp.so, csrc/graph_kernels.hip): same edge set, built on the GPU
# --------------------------------------------------------------------------------------------------
def build_edges_gpu(v, node_ptr, n_free, k1):
    """Batched counterpart of :func:`build_edges` on the GPU.  ``v`` [sumN, C] float32 (cuda),
    ``node_ptr`` [G+1] int32 (cuda), ``n_free`` / ``k1`` length-G sequences or int32 tensors.
    Returns ``(edge_index [2, sumE] int64 with graph-local ids, edge_ptr [G+1] int32)`` on the device,
    identical to coalescing each graph's kNN edges on the host."""
    import ctypes
    from . import _lib
    dev = v.device
    if dev.type != 'cuda':
        raise RuntimeError('build_edges_gpu needs device tensors; use build_edges on the host')
    G = int(node_ptr.numel() - 1)
    as_i32 = lambda x: (x if torch.is_tensor(x) else torch.tensor(list(x))).to(device=dev, dtype=torch.int32).contiguous()  # noqa: E731
    n_free_t, k1_t = as_i32(n_free), as_i32(k1)
    kmax = int(k1_t.max().item()) if G else 1
    v = v.float().contiguous()
    total = int(v.shape[0])
    b = _lib.GraphBuildBatch(G, total, max(kmax, 1), int(v.shape[1]), v.data_ptr(), node_ptr.data_ptr(),
                             n_free_t.data_ptr(), k1_t.data_ptr())
    need = ctypes.c_size_t()
    _lib.check(_lib.lib().gnnmp_graph_workspace_bytes(ctypes.byref(b), ctypes.byref(need)), 'gnnmp_graph_workspace_bytes')
    ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
    # 4 k1 N columns is the no-duplicate worst case; kNN graphs coalesce to ~1.4 k1 N, so start with 1.7 k1 N and
    # repeat with the exact size in the rare case that was not enough (the kernels never write past `cap` and
    # edge_ptr always reports the true total)
    worst = 4 * max(kmax, 1) * max(total, 1)
    cap = min(worst, int(1.7 * max(kmax, 1) * max(total, 1)) + 4096)
    edge_ptr = torch.zeros(G + 1, dtype=torch.int32, device=dev)
    while True:
        out = torch.empty((2, cap), dtype=torch.int64, device=dev)
        with torch.cuda.device(dev):
            st = torch.cuda.current_stream().cuda_stream
            _lib.check(_lib.lib().gnnmp_graph_build(ctypes.byref(b), out.data_ptr(), cap, edge_ptr.data_ptr(), ws.data_ptr(),
                                                    ws.numel(), st), 'gnnmp_graph_build')
        n_edges = int(edge_ptr[-1].item())           # one scalar read-back: the caller needs the size
        if n_edges <= cap:
            break
        cap = n_edges
    return out[:, :n_edges].contiguous(), edge_ptr


def create_data_gpu(free, collided, goal_state, k, device):
    """:func:`create_data` with the edge construction on the GPU (identical edge_index); ``v`` and
    ``edge_index`` are returned on ``device`` so the explorer can consume them without a host round trip."""
    import numpy as np
    vf = torch.tensor(np.asarray(free), dtype=torch.float32).reshape(len(free), -1)
    C = vf.shape[1]
    vc = torch.tensor(np.asarray(collided), dtype=torch.float32).reshape(-1, C)
    v = torch.cat((vf, vc), dim=0)
    labels = torch.zeros(v.shape[0], 3)
    labels[:len(free), 0] = 1
    labels[len(free):, 1] = 1
    labels[1, 2] = 1
    vd = v.to(device)
    ptr = torch.tensor([0, v.shape[0]], dtype=torch.int32, device=device)
    ei, _ = build_edges_gpu(vd, ptr, [len(free)], [k1_of(k, len(free))])
    return {'goal': torch.tensor(np.asarray(goal_state), dtype=torch.float32), 'v': v, 'labels': labels,
            'edge_index': ei, 'v_dev': vd}

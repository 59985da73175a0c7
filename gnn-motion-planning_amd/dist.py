"""Problem-level sharding across the GPUs of one node.

One planning problem = one graph = one forward, with no cross-graph term anywhere
(eval_gnn.py:113-116), so ranks never exchange data on the compute path.  The only collective is the
gather of RESULTS (edge scores, or the seven per-problem planner numbers of eval_gnn.py:120-122):
``torch.distributed`` all_gather_into_tensor on one padded buffer, i.e. RCCL over xGMI with the ``nccl``
backend on GPUs and gloo in the CPU tests.  Payloads are small (about 45 KB of scores per 1000-node
graph), so the gather is latency-bound; it is padded to the largest shard because all_gather needs
equal sizes.
"""
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world, weights=None):
    """Contiguous block [lo, hi) of ``n_items`` owned by ``rank``.  Without weights the blocks differ by
    at most one item; with per-item ``weights`` (e.g. edge counts, the cost driver) the split points
    are the weight quantiles."""
    if weights is None:
        base, rem = divmod(n_items, world)
        lo = rank * base + min(rank, rem)
        return lo, lo + base + (1 if rank < rem else 0)
    w = torch.as_tensor(weights, dtype=torch.float64)
    assert w.numel() == n_items
    cum = torch.cat((torch.zeros(1, dtype=torch.float64), w.cumsum(0)))
    total = float(cum[-1])
    cuts = [0]
    for r in range(1, world):
        target = total * r / world
        cuts.append(int(torch.searchsorted(cum, torch.tensor(target, dtype=torch.float64), right=False)))
    cuts.append(n_items)
    for i in range(1, len(cuts)):                       # keep monotone
        cuts[i] = max(cuts[i], cuts[i - 1])
    return cuts[rank], cuts[rank + 1]


def gather_variable(local, group=None):
    """all_gather of 1-D tensors of different lengths; returns the list of every rank's tensor (each on the local device).
    Two collectives on flat buffers (``all_gather_into_tensor``: one RCCL launch each, no per-rank tensor lists): the
    lengths, then ONE padded payload buffer [world, cap]; a single device-to-host read of the ``world`` lengths.  Works
    with world_size 1 without an initialised process group."""
    if not (dist.is_available() and dist.is_initialized()):
        return [local]
    world = dist.get_world_size(group)
    local = local.contiguous().reshape(-1)
    n = torch.tensor([local.numel()], dtype=torch.int64, device=local.device)
    sizes = torch.empty(world, dtype=torch.int64, device=local.device)
    dist.all_gather_into_tensor(sizes, n, group=group)
    sizes = sizes.tolist()                               # the one host read
    cap = max(sizes + [1])
    pad = torch.zeros(cap, dtype=local.dtype, device=local.device)
    pad[:local.numel()] = local
    out = torch.empty(world * cap, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    return [out[r * cap:r * cap + s] for r, s in enumerate(sizes)]


def gather_problem_results(rows, group=None):
    """``rows``: float64 tensor [n_local, k] of per-problem numbers (e.g. success, path cost, smoothed
    cost, c_explore, c_smooth, total, total_explore).  Returns [n_total, k] in rank order."""
    k = rows.shape[1] if rows.dim() == 2 else 0
    parts = gather_variable(rows.reshape(-1), group)
    return torch.cat([p.reshape(-1, k) for p in parts], dim=0) if k else torch.cat(parts)


def run_mixed(problems, models, loop=5):
    """Score a mixed set of planning problems (BASELINE configs[3]: maze / snake / ur5 / kuka together).
    Each problem is a dict with ``env`` (family key into ``models``) plus ``v, goal, obstacles, edge_index``;
    problems are bucketed per family -- one batched forward per family, since every family has its own
    (C, d, S) and weights -- and the per-edge scores come back in the caller's problem order."""
    from .batch import GraphBatch
    buckets = {}
    for i, p in enumerate(problems):
        buckets.setdefault(p['env'], []).append(i)
    out = [None] * len(problems)
    for env, idxs in buckets.items():
        m = models[env]
        dev = problems[idxs[0]]['v'].device
        b = GraphBatch.from_graphs([problems[i] for i in idxs], m.obs_size, dev)
        for i, sc in zip(idxs, b.split_edges(m.forward_batch(b, loop))):
            out[i] = sc
    return out

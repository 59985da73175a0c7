"""Problem-level sharding across the GPUs of one node.

One planning problem = one graph = one forward, with no cross-graph term anywhere
(eval_gnn.py:113-116), so ranks never exchange data on the compute path.  The only collective is the
gather of RESULTS (edge scores, or the seven per-problem planner numbers of eval_gnn.py:120-122):
``torch.distributed`` all_gather_into_tensor on one padded buffer, i.e. RCCL over xGMI with the ``nccl``
backend on GPUs and gloo in the CPU tests.  Payloads are small (about 45 KB of scores per 1000-node
graph), so the gather is latency-bound; it is padded to the largest shard because all_gather needs
equal sizes.
"""
import os
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


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE configs[3]: a MIXED set of planning problems (maze / snake / ur5 / kuka -- every family has its own (C, d, S),
# weights and kernel instantiation: constructor sites str2name.py:14,22,38,46 of the reference) sharded over the GPUs of a node.
# "mixed-env batches are first bucketed by env family so each rank runs one kernel instantiation per
# bucket".  shard_range() cuts the CALLER'S order into contiguous blocks, which on 8 ranks turns a 256-problem job into four
# tiny batches per rank; mixed_plan() buckets first and cuts the family-major order at cost quantiles, so a rank gets at most
# two families (one while no family is smaller than a rank's share and the cuts snap to a family boundary).
# ---------------------------------------------------------------------------------------------------------------------
# device time per reference-formulation FLOP relative to the d = 32 fp32 kernels, measured on one MI355X at 64 problems per
# family (profiles/r04_cfg4_mixed.txt): d = 64 in fp32 runs the non-resident pre kernel and one message workgroup per CU
_REL_TIME_PER_FLOP = {(32, 'fp32'): 1.0, (64, 'fp32'): 0.82, (32, 'bf16'): 0.45, (64, 'bf16'): 0.25,
                      (32, 'bf16x3'): 0.95, (64, 'bf16x3'): 0.9}


def forward_cost(n_nodes, n_edges, n_obs, C, d, S, loop=5, mlp_dtype='fp32'):
    """Predicted device time of ONE problem's explorer forward, in arbitrary units (only ratios matter): the FLOPs of the
    reference formulation (model.py:115-150) times the measured relative time per FLOP of the kernels
    that run this (d, operand mode).  E d^2 alone ignores that the obstacle attention of a 116-cell maze costs as much as the
    whole message passing, and that a d = 64 graph costs 2.4 x a d = 32 one at equal E."""
    N, E, O = float(n_nodes), float(n_edges), float(n_obs)
    f_enc = 2 * N * (4 * C * d + d * d) + 4 * E * (2 * C * d + d * d) + 2 * N * (C * d + d * d) + 4 * O * (S * d + d * d)
    f_att = 3 * ((N + E) * (10 * d * d + 4 * d * (O + 1)) + 16 * O * d * d)
    f_loop = loop * (8 * N * d * d + 12 * E * d * d + E * d + 4 * N * d * d) + 4 * N * d * d
    f_pol = E * (8 * d * d + 2 * d)
    return (f_enc + f_att + f_loop + f_pol) * _REL_TIME_PER_FLOP.get((d, mlp_dtype), 1.0)


def problem_costs(problems, models, loop=5):
    """forward_cost of every problem of a mixed set (``problems[i]['env']`` keys ``models``)."""
    out = []
    for p in problems:
        m = models[p['env']]
        out.append(forward_cost(p['v'].shape[0], p['edge_index'].shape[1], p['obstacles'].reshape(-1, m.obs_size).shape[0],
                                m.config_size, m.embed_size, m.obs_size, loop, getattr(m, 'mlp_dtype', 'fp32')))
    return out


def mixed_plan(families, costs, world, snap=0.08):
    """Family-aware split of a mixed problem set over ``world`` ranks.  ``families[i]`` is problem i's family key, ``costs[i]``
    its predicted time.  Problems are put in family-major order (families by decreasing total cost, caller order inside a
    family) and that sequence is cut at the cost quantiles; a cut closer than ``snap`` x (a rank's share) to a family boundary
    moves onto it, so no rank is left with a sliver of a second family.  Returns ``plan[rank]`` = list of problem indices
    (caller numbering).  Every rank computes the same plan from the same metadata: nothing is communicated."""
    n = len(families)
    assert len(costs) == n
    by_fam = {}
    for i, f in enumerate(families):
        by_fam.setdefault(f, []).append(i)
    fam_order = sorted(by_fam, key=lambda f: (-sum(costs[i] for i in by_fam[f]), str(f)))
    order = [i for f in fam_order for i in by_fam[f]]
    cum = [0.0]
    for i in order:
        cum.append(cum[-1] + float(costs[i]))
    total = cum[-1]
    bounds, pos = [], 0
    for f in fam_order[:-1]:
        pos += len(by_fam[f])
        bounds.append(pos)
    share = total / world if world else 0.0
    cuts = [0]
    for r in range(1, world):
        target = total * r / world
        # first position whose prefix cost reaches the target (bisect on the monotone prefix sums)
        lo, hi = 0, n
        while lo < hi:
            mid = (lo + hi) // 2
            if cum[mid] < target:
                lo = mid + 1
            else:
                hi = mid
        cut = lo
        if cut > 0 and target - cum[cut - 1] < cum[cut] - target:
            cut -= 1                                   # the nearer of the two neighbouring positions
        for bnd in bounds:
            if abs(cum[bnd] - target) <= snap * share:
                cut = bnd
                break
        cuts.append(max(cut, cuts[-1]))
    cuts.append(n)
    return [order[cuts[r]:cuts[r + 1]] for r in range(world)]


def shard_mixed(problems, rank, world, models, loop=5):
    """Indices (caller numbering) of the problems of a mixed set that ``rank`` of ``world`` scores: mixed_plan over
    problem_costs.  ``[problems[i] for i in shard_mixed(...)]`` goes to :class:`MixedJob` / :func:`run_mixed`."""
    return mixed_plan([p['env'] for p in problems], problem_costs(problems, models, loop), world)[rank]


class MixedJob:
    """A mixed problem set resident on one GPU, scored with every family's batched forward on its OWN stream.

    run_mixed() assembles the per-family batches on every call and runs the family forwards back to back on one stream: at
    64 problems per family each forward under-fills the device (launch tails, the serial prep -> obstacle -> pre chain), and
    the job costs the SUM of four under-filled forwards.  Here the batches, workspaces and score buffers are built once
    (inputs resident, like bench.py's headline batch) and run() forks one stream per family from the caller's stream, most
    expensive family first, and joins them: the tails of one family's launches overlap the other families' kernels.  Same
    kernels on the same inputs: scores are byte-identical to run_mixed (tests/test_full_size_mixed_gpu.py)."""

    def __init__(self, problems, models, loop=5, concurrent=True):
        from .batch import GraphBatch
        self.loop, self.models, self.n = int(loop), models, len(problems)
        buckets = {}
        for i, p in enumerate(problems):
            buckets.setdefault(p['env'], []).append(i)
        costs = problem_costs(problems, models, loop)
        self.parts = []                                  # (env, caller indices, batch, workspace, scores, stream)
        for env in sorted(buckets, key=lambda e: -sum(costs[i] for i in buckets[e])):
            idxs = buckets[env]
            m = models[env]
            dev = problems[idxs[0]]['v'].device
            b = GraphBatch.from_graphs([problems[i] for i in idxs], m.obs_size, dev)
            ws = torch.empty(m.workspace_bytes(b), dtype=torch.uint8, device=dev)
            out = torch.empty(max(b.total_edges, 1), dtype=torch.float32, device=dev)
            # (a higher hardware-queue priority for the most expensive family's stream was measured and dropped: 59.4 k -> 57.1 k graphs/s,
            # profiles/r05_cfg4_mixed.txt)
            self.parts.append((env, idxs, b, ws, out, torch.cuda.Stream(dev) if concurrent else None))

    def run(self):
        """One forward per family; returns the per-problem score tensors in the caller's order (views into the job's own
        buffers: valid until the next run())."""
        res = [None] * self.n
        # the caller's stream ON THE DEVICE THE PROBLEMS LIVE ON (not of whatever device is current)
        cur = torch.cuda.current_stream(self.parts[0][2].v.device) if self.parts else None
        for env, idxs, b, ws, out, st in self.parts:
            if st is None:
                sc = self.models[env].forward_batch(b, self.loop, ws=ws, out=out)
            else:
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    sc = self.models[env].forward_batch(b, self.loop, ws=ws, out=out)
            for i, s in zip(idxs, b.split_edges(sc)):
                res[i] = s
        for part in self.parts:
            if part[5] is not None:
                cur.wait_stream(part[5])
        return res

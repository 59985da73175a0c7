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
# Cost model of one family's batched forward on one MI355X: device milliseconds as a function of the batch's reference-formulation
# GFLOP (reference_flops below), one piecewise-linear curve per (d, operand mode) through the points of tools/cost_sweep.py (per
# family at 1 ... 128 problems, 1000-node k1 = 8, loop 5; families of one (d, mode) pooled per batch size;
# profiles/r06_cost_sweep.txt).  The curve does not go through the origin -- a forward costs 0.1-0.5 ms before it has any width (the
# serial prep -> obstacle -> pre -> 5 x message -> policy launch chain and its under-filled tails) -- and it is concave below ~32
# problems.  The round-5 model (FLOPs x a constant) had neither: it predicted slowest / mean 1.018 for the 8-way split of the
# 256-problem configs[3] job where the measured shards gave 1.263, because three 17-problem kuka7 shards each paid the fixed part again.
_COST_CURVE = {
    (32, 'fp32'): ((0.0, 0.122), (1.7, 0.136), (3.4, 0.154), (6.8, 0.166), (13.7, 0.224), (27.3, 0.341), (54.6, 0.520), (109.3, 0.845), (218.7, 1.532)),
    (64, 'fp32'): ((0.0, 0.357), (5.7, 0.397), (11.3, 0.408), (22.6, 0.432), (45.2, 0.521), (90.4, 0.855), (180.8, 1.456), (361.8, 2.110), (724.0, 3.885)),
    (32, 'bf16'): ((0.0, 0.077), (1.7, 0.086), (3.4, 0.097), (6.8, 0.100), (13.7, 0.120), (27.3, 0.163), (54.6, 0.215), (109.3, 0.325), (218.7, 0.561)),
    (64, 'bf16'): ((0.0, 0.110), (5.7, 0.122), (11.3, 0.128), (22.6, 0.149), (45.2, 0.165), (90.4, 0.235), (180.8, 0.369), (361.8, 0.523), (724.0, 0.900)),
}
_COST_CURVE[(32, 'bf16x3')] = tuple((g, 0.94 * t) for g, t in _COST_CURVE[(32, 'fp32')])       # not swept: the measured bf16x3 / fp32 step ratios
_COST_CURVE[(64, 'bf16x3')] = tuple((g, 1.05 * t) for g, t in _COST_CURVE[(64, 'fp32')])


def reference_flops(n_nodes, n_edges, n_obs, C, d, S, loop=5):
    """FLOPs of the reference formulation of one problem's explorer forward (model.py:115-150)."""
    N, E, O = float(n_nodes), float(n_edges), float(n_obs)
    f_enc = 2 * N * (4 * C * d + d * d) + 4 * E * (2 * C * d + d * d) + 2 * N * (C * d + d * d) + 4 * O * (S * d + d * d)
    f_att = 3 * ((N + E) * (10 * d * d + 4 * d * (O + 1)) + 16 * O * d * d)
    f_loop = loop * (8 * N * d * d + 12 * E * d * d + E * d + 4 * N * d * d) + 4 * N * d * d
    f_pol = E * (8 * d * d + 2 * d)
    return f_enc + f_att + f_loop + f_pol


def forward_cost(n_nodes, n_edges, n_obs, C, d, S, loop=5, mlp_dtype='fp32'):
    """Work of ONE problem's explorer forward in reference-formulation GFLOP -- the unit :func:`batch_time` and :func:`mixed_plan`
    count in.  (E d^2 alone would ignore that the obstacle attention of a 116-cell maze costs as much as the whole message passing.)
    ``mlp_dtype`` is accepted for symmetry with batch_time and does not change the count."""
    return reference_flops(n_nodes, n_edges, n_obs, C, d, S, loop) / 1e9


def batch_time(d, mlp_dtype, gflop):
    """Predicted device milliseconds of ONE batched forward of a (d, operand mode) family over problems worth ``gflop`` in total:
    the measured curve, linear between its points and beyond the last one; 0 for an empty batch."""
    if gflop <= 0:
        return 0.0
    pts = _COST_CURVE.get((d, mlp_dtype), _COST_CURVE[(32, 'fp32')])
    for (g0, t0), (g1, t1) in zip(pts, pts[1:]):
        if gflop <= g1:
            return t0 + (t1 - t0) * (gflop - g0) / (g1 - g0)
    (g0, t0), (g1, t1) = pts[-2], pts[-1]
    return t1 + (t1 - t0) / (g1 - g0) * (gflop - g1)


def problem_costs(problems, models, loop=5):
    """forward_cost of every problem of a mixed set (``problems[i]['env']`` keys ``models``)."""
    out = []
    for p in problems:
        m = models[p['env']]
        out.append(forward_cost(p['v'].shape[0], p['edge_index'].shape[1], p['obstacles'].reshape(-1, m.obs_size).shape[0],
                                m.config_size, m.embed_size, m.obs_size, loop, getattr(m, 'mlp_dtype', 'fp32')))
    return out


def plan_times(plan, families, costs, time_of=None):
    """Predicted milliseconds of every rank of ``plan``: for each family the rank holds, ``time_of[family]`` of the summed costs of
    its problems of that family (default: the sum itself), added up over the families."""
    out = []
    for r in plan:
        per = {}
        for i in r:
            per[families[i]] = per.get(families[i], 0.0) + float(costs[i])
        out.append(sum((time_of[f](g) if time_of and f in time_of else g) for f, g in per.items()))
    return out


def mixed_plan(families, costs, world, time_of=None):
    """Family-aware split of a mixed problem set over ``world`` ranks.  ``families[i]`` is problem i's family key, ``costs[i]`` its
    work (:func:`forward_cost`: GFLOP), ``time_of[family]`` a monotone function from the summed work of one rank's problems of that
    family to the milliseconds of their batched forward (:func:`family_time_curves`; default: identity, i.e. plain cost shares).
    Problems are put in family-major order (families by decreasing total time, caller order inside a family) and that sequence is
    cut into at most ``world`` contiguous blocks so that the SLOWEST block's predicted time -- the sum over the families it touches
    of time_of(work it holds of that family) -- is as small as it can be (bisection on the bound, greedy fill: the classic linear
    partition with a block cost that knows a forward's fixed part).  A family too small to be worth a rank of its own therefore
    shares one (MixedJob runs the families of a rank on their own streams) instead of being sliced thinner, and no rank is handed a
    sliver of a second family that costs more in fixed time than it relieves.  Neighbouring blocks are then evened out below that
    bound.  Greedy at the optimal bound may need fewer than ``world`` blocks: the remaining ranks stay idle (a further cut could not
    lower the slowest block).  Returns ``plan[rank]`` =
    list of problem indices (caller numbering).  Every rank computes the same plan from the same metadata: nothing is communicated."""
    n = len(families)
    assert len(costs) == n
    ident = lambda g: g                                                                                      # noqa: E731
    tf = lambda f: (time_of.get(f, ident) if time_of else ident)                                             # noqa: E731
    by_fam = {}
    for i, f in enumerate(families):
        by_fam.setdefault(f, []).append(i)
    fam_order = sorted(by_fam, key=lambda f: (-tf(f)(sum(float(costs[i]) for i in by_fam[f])), str(f)))
    order = [i for f in fam_order for i in by_fam[f]]
    if world <= 1 or n == 0:
        return [order] + [[] for _ in range(max(world, 1) - 1)]
    c = [float(costs[i]) for i in order]
    fam = [families[i] for i in order]
    fn = [tf(f) for f in fam]

    def fill(bound):
        """Greedy cuts under ``bound``: every block as long as its predicted time allows.  None if more than ``world`` are needed."""
        cuts, start = [0], 0
        done, run = 0.0, 0.0           # time of the block's completed families; work of its current (last) family
        for k in range(n):
            if k > start and fam[k] != fam[k - 1]:
                done, run = done + fn[k - 1](run), 0.0
            if done + fn[k](run + c[k]) > bound:
                if k == start:
                    return None                                # a single problem already exceeds the bound
                cuts.append(k)
                if len(cuts) > world:
                    return None
                start, done, run = k, 0.0, 0.0
                if fn[k](c[k]) > bound:
                    return None
            run += c[k]
        return cuts + [n]

    lo = max(f_(ck) for f_, ck in zip(fn, c))
    hi = sum(tf(f)(sum(float(costs[i]) for i in by_fam[f])) for f in fam_order)
    if fill(lo) is not None:
        hi = lo
    for _ in range(60):
        if hi - lo <= 1e-9 * hi:
            break
        mid = 0.5 * (lo + hi)
        if fill(mid) is None:
            lo = mid
        else:
            hi = mid
    cuts = fill(hi)

    def btime(a, b):
        t, run = 0.0, 0.0
        for k in range(a, b):
            if k > a and fam[k] != fam[k - 1]:
                t, run = t + fn[k - 1](run), 0.0
            run += c[k]
        return t + (fn[b - 1](run) if b > a else 0.0)
    # greedy fills the early blocks to the bound and leaves the last one whatever remains: even neighbouring blocks out (the cut
    # between two adjacent blocks moves to where the slower of the two is fastest; never raises the slowest block of the plan)
    for _ in range(4):
        moved = False
        for q in range(len(cuts) - 2, 0, -1):
            a, m, b = cuts[q - 1], cuts[q], cuts[q + 1]
            if b - a < 2:
                continue
            best, best_t = m, max(btime(a, m), btime(m, b))
            lo_k, hi_k = a + 1, b - 1
            while lo_k <= hi_k:                               # left block grows with k, right block shrinks: bisect on the crossing
                k = (lo_k + hi_k) // 2
                tl, tr = btime(a, k), btime(k, b)
                if max(tl, tr) < best_t - 1e-12:
                    best, best_t = k, max(tl, tr)
                if tl < tr:
                    lo_k = k + 1
                else:
                    hi_k = k - 1
            if best != m:
                cuts[q], moved = best, True
        if not moved:
            break
    plan = [order[cuts[i]:cuts[i + 1]] for i in range(len(cuts) - 1) if cuts[i + 1] > cuts[i]]
    return plan + [[] for _ in range(world - len(plan))]


def family_time_curves(models):
    """{family: work (GFLOP) -> batched-forward milliseconds} for the models of a mixed job (:func:`batch_time`)."""
    return {env: (lambda g, d=m.embed_size, t=getattr(m, 'mlp_dtype', 'fp32'): batch_time(d, t, g)) for env, m in models.items()}


def shard_mixed(problems, rank, world, models, loop=5):
    """Indices (caller numbering) of the problems of a mixed set that ``rank`` of ``world`` scores: mixed_plan over
    problem_costs and the families' time curves.  ``[problems[i] for i in shard_mixed(...)]`` goes to :class:`MixedJob` /
    :func:`run_mixed`."""
    return mixed_plan([p['env'] for p in problems], problem_costs(problems, models, loop), world, family_time_curves(models))[rank]


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

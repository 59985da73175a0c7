"""Host-side planner around the GPU models: the build's counterparts of the reference callers of the
hot path (behavioural spec Appendix G):

  * ``obs_data``                 eval_gnn.py:25-36  /  smoother.py:52-64
  * ``explore``                  eval_gnn.py:168-276  (greedy best-edge expansion on the explorer's output)
  * ``smooth_step`` / ``model_smooth``   smoother.py:194-216 / :233-246
  * ``eval_gnn``                 eval_gnn.py:96-145   (aggregates)

The sequential control flow and the collision checks stay on the host CPU (north_star); the GPU
models are called through the same keyword signatures the reference uses.  ``explore`` reproduces the
reference's observable behaviour including two quirks (SURVEY.md finding 0.6): the explored-edge mask is
applied with LEGACY tuple-index semantics, and the pair list is reshaped (2, -1) rather than transposed.
"""
import time
from copy import deepcopy

import numpy as np
import torch

from .graph_build import create_data, create_data_gpu


def path_cost(path):
    """Sum of segment lengths (eval_gnn.py:53-58)."""
    p = np.array(path)
    return float(sum(np.linalg.norm(p[i + 1] - p[i]) for i in range(len(p) - 1)))


def obs_data(env, free, collided, device, for_smoother=False, obstacles_only=False):
    """Tensors handed to the models next to the graph (explorer: eval_gnn.py:25-36, collided truncated
    to len(free); smoother: smoother.py:52-64, empty lists replaced by one zero row -- appended to the
    CALLER's list like the reference does -- and both truncated to 500).
    ``obstacles_only``: just the obstacle tensor.  The explorer's forward accepts ``free`` / ``collided`` and never reads them
    (model.py:115: they are not used anywhere in the body), so the host loop need not build two [n, C] tensors from lists of
    rows and copy them to the device before every forward (0.15 ms of the 0.9 ms span at n = 500)."""
    if obstacles_only:
        return {'obstacles': torch.tensor(np.asarray(env.obstacles), dtype=torch.float32).to(device)}
    if for_smoother:
        if not len(free):
            free.append([0. for _ in range(env.config_dim)])
        if not len(collided):
            collided.append([0. for _ in range(env.config_dim)])
        free, collided = free[:500], collided[:500]
        coll_t = torch.tensor(np.array(collided), dtype=torch.float32)
    else:
        coll_t = torch.tensor(np.array(collided), dtype=torch.float32)[:len(free)]
    return {'free': torch.tensor(np.array(free), dtype=torch.float32).to(device), 'collided': coll_t.to(device),
            'obstacles': torch.tensor(np.asarray(env.obstacles), dtype=torch.float32).to(device)}


def _mask_policy(P, labels, explored, explored_edges):
    """eval_gnn.py:198-202 on a dense numpy [N, N] matrix P[target, source]."""
    n = P.shape[0]
    P[np.arange(n), np.arange(n)] = 0
    P[:, explored] = 0
    coll = labels[:, 1] == 1
    P[:, coll] = 0
    P[coll, :] = 0
    idx = np.array(explored_edges).reshape(2, -1)     # reshape, not transpose: part of the observable behaviour
    P[idx[0], idx[1]] = 0                             # legacy "sequence of sequences == tuple" indexing
    return P


def greedy_expand(P, v, env, state):
    """The inner while-loop of ``explore`` (eval_gnn.py:204-233) on a masked dense policy.
    ``state``: dict(explored, explored_edges, costs, prev); mutated in place.  Returns the node-index
    path if the goal region was reached, else None."""
    explored, explored_edges = state['explored'], state['explored_edges']
    while True:
        sub = P[explored, :]
        if not (sub.sum(dtype=np.float32) != 0):      # float sum test of the reference (eval_gnn.py:204)
            return None
        rows, cols = np.nonzero(sub)                  # row-major: (explored order, column)
        a = int(np.argmax(sub[rows, cols]))           # first maximum
        end_a, end_b = int(explored[rows[a]]), int(cols[a])
        explored_edges.extend([[end_a, end_b], [end_b, end_a]])
        if env._edge_fp(v[end_a], v[end_b]):
            explored.append(end_b)
            state['costs'][end_b] = state['costs'][end_a] + np.linalg.norm(v[end_a] - v[end_b])
            state['prev'][end_b] = end_a
            P[:, end_b] = 0
            if env.in_goal_region(v[end_b]):
                path, node = [end_b], end_b
                while node != 0:
                    node = state['prev'][node]
                    path.append(node)
                path.reverse()
                return path
        else:
            P[end_a, end_b] = 0
            P[end_b, end_a] = 0


def greedy_expand_sparse(scores, edge_index, labels, v, env, state):
    """Same decisions as ``_mask_policy`` + ``greedy_expand`` but on the per-edge scores (E numbers
    instead of the dense N x N matrix): a max-heap over the live cells
    ``P[a, b]`` with ``a`` explored, keyed so that ties break exactly like the dense row-major argmax
    (position of ``a`` in the explored list, then column ``b``).  ``scores[e]`` is the score of column e
    of ``edge_index`` = dense cell ``P[target, source]``.
    One deliberate difference: the reference's loop condition is ``policy[explored, :].sum() != 0``
    (eval_gnn.py:204), so it also stops when live cells remain but their float32 sum cancels to exactly 0;
    this frontier (and the device one, maze_kernels.hip) stops only when no live cell remains.  For real
    network scores the two differ with probability ~0 (never on the 1000 recorded problems)."""
    import heapq
    explored, explored_edges = state['explored'], state['explored_edges']
    src, dst = edge_index[0], edge_index[1]
    coll = labels[:, 1] == 1
    rows = {}                                            # target a -> {source b: score}
    for e in range(src.shape[0]):
        a, b, sc = int(dst[e]), int(src[e]), scores[e]
        if a == b or sc == 0 or coll[a] or coll[b]:
            continue
        rows.setdefault(a, {})[b] = sc                   # later duplicate columns overwrite, like index_put
    idx = np.array(explored_edges).reshape(2, -1)        # the reference's quirk (finding 0.6)
    for a, b in zip(idx[0].tolist(), idx[1].tolist()):
        if a in rows:
            rows[a].pop(b, None)
    is_explored = np.zeros(labels.shape[0], dtype=bool)
    is_explored[explored] = True
    heap = []

    def push_row(pos, a):
        for b, sc in rows.get(a, {}).items():
            if not is_explored[b]:
                heapq.heappush(heap, (-float(sc), pos, b, a))

    for pos, a in enumerate(explored):
        push_row(pos, a)
    while heap:
        neg, pos, end_b, end_a = heapq.heappop(heap)
        live = rows.get(end_a, {})
        if is_explored[end_b] or live.get(end_b) is None or float(live[end_b]) != -neg:
            continue                                      # cell was zeroed since it was pushed
        explored_edges.extend([[end_a, end_b], [end_b, end_a]])
        if env._edge_fp(v[end_a], v[end_b]):
            explored.append(end_b)
            is_explored[end_b] = True
            state['costs'][end_b] = state['costs'][end_a] + np.linalg.norm(v[end_a] - v[end_b])
            state['prev'][end_b] = end_a
            if env.in_goal_region(v[end_b]):
                path, node = [end_b], end_b
                while node != 0:
                    node = state['prev'][node]
                    path.append(node)
                path.reverse()
                return path
            push_row(len(explored) - 1, end_b)
        else:
            live.pop(end_b, None)
            rows.get(end_b, {}).pop(end_a, None)
    return None


def _steer_round(path, target, env):
    """One sweep over the interior waypoints, left to right: each moves at most env.RRT_EPS towards its target and the move
    is kept only if the edges to both neighbours stay free -- the left neighbour already at its new place, the right one
    still at its old one.  Returns the new path and the summed distance still to go of the waypoints that moved."""
    out = deepcopy(path)
    remaining = 0
    for i in range(1, len(path) - 1):
        gap = np.linalg.norm(path[i] - target[i])
        cand = target[i] if gap < env.RRT_EPS else env.interpolate(path[i], target[i], env.RRT_EPS / gap)
        out[i] = cand
        if env._edge_fp(out[i - 1], cand) and env._edge_fp(out[i + 1], cand):
            remaining += np.linalg.norm(cand - target[i])
        else:
            out[i] = path[i]
    return out, remaining


def smooth_step(old_path, new_path, env):
    """Collision-checked steering of the smoothing stage (``proposed_path_smootherv2``, smoother.py:194-216): as many
    sweeps as the farthest waypoint needs RRT_EPS steps, stopping early once every accepted move has arrived."""
    far = np.linalg.norm(np.array(old_path) - np.array(new_path), axis=-1).max()
    path = deepcopy(old_path)
    for _ in range(int(np.ceil(far / env.RRT_EPS))):
        path, remaining = _steer_round(path, new_path, env)
        if remaining < 1e-5:
            break
    return path


def chain_edge_index(n):
    """i <-> i+1 in both directions, then self loops appended (smoother.py:238-241)."""
    a, b = torch.arange(1, n), torch.arange(0, n - 1)
    e = torch.cat((torch.stack((a, b)), torch.stack((b, a))), dim=1)
    loops = torch.arange(n)
    return torch.cat((e, torch.stack((loops, loops))), dim=1)


def _chain_edge_indices(lengths):
    """:func:`chain_edge_index` of every path length in ``lengths``, side by side ([2, sum (3 n - 2)] int64, path-local ids):
    one numpy pass instead of seven torch calls per path."""
    n = np.asarray(lengths, dtype=np.int64)
    per = 3 * n - 2
    off = np.zeros(len(n) + 1, dtype=np.int64)
    off[1:] = np.cumsum(per)
    j = np.arange(off[-1], dtype=np.int64) - np.repeat(off[:-1], per)          # position inside the path's block
    nn = np.repeat(n, per)
    out = np.empty((2, off[-1]), dtype=np.int64)
    fwd, bwd = j < nn - 1, (j >= nn - 1) & (j < 2 * nn - 2)
    loop = j - (2 * nn - 2)
    out[0] = np.where(fwd, j + 1, np.where(bwd, j - (nn - 1), loop))
    out[1] = np.where(fwd, j, np.where(bwd, j - (nn - 1) + 1, loop))
    return out


@torch.no_grad()
def model_smooth(model, free, collided, old_path, env, device, iters=5, trace=None):
    """smoother.py:233-246: 5 x (smoother forward with loop=1 -> collision-checked steering)."""
    for _ in range(iters):
        data = obs_data(env, free, collided, device, for_smoother=True)
        path_t = torch.tensor(np.array(old_path), dtype=torch.float32).to(device)
        new_path = model(path=path_t, edge_index=chain_edge_index(len(old_path)).to(device), loop=1, **data)
        new_path = new_path.detach().cpu().numpy()
        if trace is not None:
            trace.append((np.array(old_path, dtype=np.float32), new_path.copy()))
        old_path = smooth_step(old_path, new_path, env)
    return old_path


@torch.no_grad()
def explore(env, model, model_s, smooth=True, batch=500, t_max=1000, k=30, smoother='model', loop=5, device='cuda',
            trace=None, sparse=False, gpu_graph=False, reference_kwargs=None):
    """Counterpart of ``explore`` (eval_gnn.py:168-276).  Returns the same result dict.
    ``reference_kwargs=True`` hands the model every keyword the reference does (``free``, ``collided``, ``labels``: built, copied to
    the device and then ignored by the forward, model.py:115); ``False`` passes only what the forward reads -- same output.  Default
    (None): False for this package's ``EncoderProcessDecoder``, True for any other model (one with the reference's signature needs them).
    ``sparse=True`` asks the model for per-edge scores (``edge_scores``) and runs the heap-based
    frontier instead of pulling the dense N x N matrix to the host; decisions are identical.
    ``gpu_graph=True`` builds the kNN graph on the device (graph_kernels.hip; same edge_index)."""
    if reference_kwargs is None:
        from .explorer import EncoderProcessDecoder
        reference_kwargs = not isinstance(model, EncoderProcessDecoder)
    c0 = env.collision_check_count
    t0 = time.time()
    forward = 0.
    success, path, smooth_path = False, [], []
    free, collided = env.sample_n_points(batch, need_negative=True)
    collided = collided[:len(free)]
    free = [env.init_state] + [env.goal_state] + list(free)
    state = {'explored': [0], 'explored_edges': [[0, 0]], 'costs': {0: 0.}, 'prev': {0: 0}}
    # what `forward` (the span eval_gnn.py:193-196 times) is made of, seconds: obs_data (host tensors + their copies), h2d (graph
    # tensors), module_call (host time of model(**kw): enqueue only, nothing waits), d2h_wait (.cpu(): kernels finishing + the copy)
    split = {'obs_data': 0., 'h2d': 0., 'module_call': 0., 'd2h_wait': 0., 'calls': 0}
    make_data = (lambda f, c: create_data_gpu(f, c, env.goal_state, k, device)) if gpu_graph else \
        (lambda f, c: create_data(f, c, env.goal_state, k))
    data = make_data(free, collided)
    while not success and (len(free) - 2) <= t_max:
        t1 = time.time()
        od = obs_data(env, free, collided, device, obstacles_only=not reference_kwargs)
        t2 = time.time()
        kw = dict(goal=data['goal'].to(device), v=data.get('v_dev', data['v']).to(device),
                  edge_index=data['edge_index'].to(device), loop=loop, **od)
        if reference_kwargs:
            kw['labels'] = data['labels'].to(device)
        t3 = time.time()
        ei = data['edge_index'].cpu().numpy()
        v = data['v'].numpy()
        split['obs_data'] += t2 - t1
        split['h2d'] += t3 - t2
        split['calls'] += 1
        if sparse:
            t4 = time.time()
            sc = model.edge_scores(**kw)
            t5 = time.time()
            sc = sc.detach().cpu().numpy()                           # E floats instead of N^2
            forward += time.time() - t1
            split['module_call'] += t5 - t4
            split['d2h_wait'] += time.time() - t5
            if trace is not None:
                trace.setdefault('forwards', []).append({'v': v.copy(), 'edge_index': ei.copy(), 'scores': sc.copy()})
            found = greedy_expand_sparse(sc, ei, data['labels'].numpy(), v, env, state)
        else:
            t4 = time.time()
            P = model(**kw)
            t5 = time.time()
            P = P.detach().cpu().numpy()                             # the implicit sync of eval_gnn.py:195
            forward += time.time() - t1
            split['module_call'] += t5 - t4
            split['d2h_wait'] += time.time() - t5
            if trace is not None:
                trace.setdefault('forwards', []).append({'v': v.copy(), 'edge_index': ei.copy(),
                                                         'scores': P[ei[1], ei[0]].copy()})
            P = _mask_policy(P, data['labels'].numpy(), state['explored'], state['explored_edges'])
            found = greedy_expand(P, v, env, state)
        if found is not None:
            success, path = True, found
        if not success:
            if not smooth:
                return []
            if (batch + len(free) - 2) > t_max:
                break
            new_free, new_coll = env.sample_n_points(batch, need_negative=True)
            free = free + list(new_free)
            collided = (collided + list(new_coll))[:len(free)]
            data = make_data(free, collided)
    c_explore = env.collision_check_count - c0
    c1 = env.collision_check_count
    t1 = time.time()
    if success and smooth:
        path = list(data['v'][path].numpy())
        if smoother == 'model':
            smooth_path = model_smooth(model_s, free, collided, path, env, device,
                                       trace=None if trace is None else trace.setdefault('smooth', []))
        else:
            smooth_path = path
    c_smooth = env.collision_check_count - c1
    if not smooth:
        return list(data['v'][path].numpy()), free, collided
    return {'c_explore': c_explore, 'c_smooth': c_smooth, 'data': data, 'explored': state['explored'],
            'forward': forward, 'forward_split': split, 'total': time.time() - t0, 'total_explore': t1 - t0, 'success': success, 't0': t0,
            'path': path, 'smooth_path': smooth_path, 'explored_edges': state['explored_edges']}


def eval_gnn(env, indexes, model, model_s, seed=1234, smooth=True, batch=500, t_max=500, k=30, device='cuda', **kw):
    """Counterpart of ``eval_gnn`` (eval_gnn.py:96-145): the seven per-problem numbers and their
    aggregates, same order as the reference's return tuple."""
    model.eval()                       # eval_gnn.py:109-110
    if model_s is not None:
        model_s.eval()
    from .hostenv import warn_if_oversubscribed
    warn_if_oversubscribed()           # a torch pool larger than the CPU quota stalls this one-core loop (hostenv.py)
    np.random.seed(seed)
    torch.manual_seed(seed)
    sol, paths, smooth_paths = [], [], []
    for index in indexes:
        env.init_new_problem(index)
        r = explore(env, model, model_s, smooth, batch=batch, t_max=t_max, k=k, device=device, **kw)
        paths.append(r['path'])
        smooth_paths.append(r['smooth_path'])
        sol.append((r['success'], path_cost(r['path']), path_cost(r['smooth_path']), r['c_explore'], r['c_smooth'],
                    r['total'], r['total_explore']))
    n_success = sum(s[0] for s in sol)
    collision_explore = float(np.mean([s[3] for s in sol]))
    collision = float(np.mean([s[3] + s[4] for s in sol]))
    running_time = float(sum(s[5] for s in sol if s[0])) / max(n_success, 1)
    solution_cost = float(sum(s[2] for s in sol if s[0])) / max(n_success, 1)
    total_time = sum(s[5] for s in sol)
    total_time_explore = sum(s[6] for s in sol)
    return (n_success, collision, running_time, solution_cost, total_time, paths, smooth_paths, collision_explore,
            total_time_explore)


def _path_cost_rows(path):
    """:func:`path_cost` of a float32 [P, dim] array without a Python loop over the segments: per-segment sqrt(x . x) in
    float32, accumulated left to right (cumsum) like the reference's running sum -- the same bits (tests/test_planner_host.py).  (27 k np.linalg.norm calls per 1024 maze
    problems were a fifth of the device planner's host time.)"""
    p = np.asarray(path)
    if p.ndim != 2 or p.shape[0] < 2:
        return 0.0
    if p.dtype != np.float32 or p.shape[1] != 2:
        # float64 rows, and rows of three or more coordinates (maze3): np.linalg.norm's BLAS dot and einsum round
        # differently in the last bit; only the 2-D float32 rows of the device planner are proven equal -- loop otherwise
        return path_cost(p)
    d = p[1:] - p[:-1]
    return float(np.cumsum(np.sqrt(np.einsum('ij,ij->i', d, d)), dtype=_COST_ACC)[-1])


# dtype the reference's running sum `0 + float32 + float32 ...` (eval_gnn.py:53-58) accumulates in under the installed NumPy:
# float32 under NEP 50 (NumPy >= 2: the Python 0 is weak), whatever older promotion rules give otherwise
_COST_ACC = np.asarray(sum([np.float32(1.0)])).dtype


def _collect(res, wall, t_explore, sol, paths, smooth_paths, rows_out):
    for r in res:
        paths.append(r['path'] if r['success'] else [])
        smooth_paths.append(r['smooth_path'] if r['success'] else [])
        sol.append((r['success'], _path_cost_rows(paths[-1]), _path_cost_rows(smooth_paths[-1]), r['c_explore'], r['c_smooth'],
                    wall, t_explore))
        if rows_out is not None:
            rows_out.append(sol[-1][:5] + (len(paths[-1]), len(r['explored'])))


def skip_maze_sampling(env, indexes, batch=500):
    """Advance the global numpy RNG exactly as the planner's sampling of the problems ``indexes`` would (nothing
    else in the default single-forward planner draws random numbers): lets rank r of a sharded evaluation start its
    block at the stream position the sequential reference loop would have reached (0.07 ms per skipped problem)."""
    from .maze2d import AttemptStream, Maze2D
    stream = AttemptStream()
    for i in indexes:
        e = Maze2D(np.asarray(env.maps[i])[None], np.asarray(env.init_states[i])[None], np.asarray(env.goal_states[i])[None])
        e.init_new_problem(0)
        e.sample_n_points_stream(stream, batch)
    stream.close()


def _pass_spans(n, chunk):
    """[lo, hi) problem ranges of the device passes of an evaluation of ``n`` problems: passes of ``chunk`` problems behind a short
    ramp (chunk / 4, chunk / 2) -- nothing overlaps the sampling of the first pass, so it is kept small (44 of 250 ms at 1024
    problems in passes of 512).  The spans partition range(n) in order."""
    chunk = max(int(chunk), 1)
    spans, c0 = [], 0
    for size in (max(chunk // 4, 1), max(chunk // 2, 1)):
        if n - c0 > chunk:
            spans.append((c0, c0 + size))
            c0 += size
    while c0 < n:
        spans.append((c0, min(c0 + chunk, n)))
        c0 += chunk
    return spans


_WORKER_STREAMS = {}


def _worker_stream(dev, i):
    """Stream of device-pass worker ``i`` on ``dev``, created once per process: torch's caching allocator keeps a pool per
    stream (and the modules a workspace per stream), so fresh streams per call would allocate the whole working set -- GBs
    at 512 problems per pass -- again every time."""
    key = (str(dev), i)
    if key not in _WORKER_STREAMS:
        _WORKER_STREAMS[key] = torch.cuda.Stream(dev)
    return _WORKER_STREAMS[key]


def eval_gnn_device(env, indexes, model, model_s, seed=1234, batch=500, k=30, device='cuda', loop=5, chunk=128,
                    rows_out=None, shard=None, workers=2, device_sampling=True):
    """:func:`eval_gnn` for 2-D maze environments with the planner itself on the device
    (:func:`explore_maze_batch`, ``chunk`` problems per device pass): same return tuple as ``eval_gnn``
    (eval_gnn.py:96-145), same per-problem decisions and collision-check counts as the one-by-one loop at the
    reference's default configuration (smoothing on, batch == t_max: one explorer forward per problem).
    The time entries are the batch wall time spread evenly over the problems of a chunk.  ``rows_out``: optional
    list that receives one (success, path cost, smoothed cost, c_explore, c_smooth, path length, explored) per problem.
    ``shard = (rank, world)``: evaluate only this rank's contiguous block of ``indexes`` (``dist.shard_range``) after
    skipping the sampling of the blocks before it, so the union over ranks equals the sequential run problem by
    problem; the aggregates returned are those of the local block (gather with ``dist.gather_problem_results``).
    ``workers``: host threads that run device passes, each on its OWN stream (bound to the worker thread, not to the pass number): the host
    part of a pass (index tables, result rows, the blocking copies between the stages) leaves the GPU idle for ~40 % of a
    pass, and a second pass in flight fills that; the chunks are independent and their results are collected in order,
    so every per-problem number is the same for any ``workers``.  Measured at 1024 problems of the published setting
    (tools/diag/planner_chunks.py, problems/s, median of 3): workers 1 / 2 / 3 at chunk 512: 5.5 k / 5.1 k / 6.1 k, chunk 256:
    5.1 k / 5.8 k / 6.5 k, chunk 128: 4.7 k / 7.1 k / 6.7 k -- small passes start the pipeline early and keep the per-stream
    pools of the caching allocator small.  Memory: the defaults (passes of 128 on two workers) reserve about 45 GB in the caching
    allocator (55 GB at three workers) -- sized for the 288 GB of an MI355X; pass ``workers=1`` and a smaller ``chunk`` elsewhere."""
    model.eval()                       # eval_gnn.py:109-110
    if model_s is not None:
        model_s.eval()
    np.random.seed(seed)
    torch.manual_seed(seed)
    indexes = list(indexes)
    if shard is not None:
        from .dist import shard_range
        lo, hi = shard_range(len(indexes), shard[0], shard[1])
        skip_maze_sampling(env, indexes[:lo], batch)
        indexes = indexes[lo:hi]
    sol, paths, smooth_paths = [], [], []
    from concurrent.futures import ThreadPoolExecutor
    dev = torch.device(device)
    if dev.type == 'cuda' and dev.index is None:
        dev = torch.device('cuda', torch.cuda.current_device())
    workers = max(1, int(workers))
    # the weight handles are built here, once, not by whichever worker comes first
    model._native(dev)
    if model_s is not None:
        model_s._native(dev)

    def prepare(span):
        pr = [dict(map=env.maps[i], init_state=env.init_states[i], goal_state=env.goal_states[i])
              for i in indexes[span[0]:span[1]]]
        # device_sampling: only the uniform draws come from the host (numpy's global generator, this one thread); classification,
        # n-th-free search and the node rows are the device's (gnnmp_maze_sample) -- same samples, same check counts, same stream
        # position afterwards as the host sampler (tests/test_maze_sample_gpu.py)
        return pr, (sample_maze_problems_device(pr, batch, k, dev) if device_sampling else sample_maze_problems(pr, batch, k))

    # INVARIANT: one stream -- and with it one workspace of each module (EncoderProcessDecoder / ModelSmoother._workspace key
    # on (device, stream)) -- per concurrently running forward.  A stream therefore belongs to a worker THREAD (bound once, in
    # the pool's initializer), never to a pass number: with passes handed out as `ci % workers`, pass i + workers could start
    # on pass i's stream while pass i was still running whenever pass i + 1 finished first, and two gnnmp_*_forward calls
    # (ctypes releases the interpreter lock) would interleave their launches on one stream and one workspace.
    import itertools
    import threading
    streams = [_worker_stream(dev, i) for i in range(workers)] if workers > 1 else [None]
    bound, next_stream = threading.local(), itertools.count()

    def bind_stream():
        bound.stream = streams[next(next_stream) % workers]          # the pool starts at most `workers` threads: one stream each

    def device_pass(ci, sampled):
        problems, pre = sampled.result()
        tm = {}
        t_pass = time.perf_counter()
        st = getattr(bound, 'stream', None)
        if st is None:
            res = explore_maze_batch(problems, model, device, batch=batch, k=k, loop=loop, model_s=model_s, timings=tm,
                                     presampled=pre)
        else:
            with torch.cuda.device(dev), torch.cuda.stream(st):
                res = explore_maze_batch(problems, model, device, batch=batch, k=k, loop=loop, model_s=model_s,
                                         timings=tm, presampled=pre)
                torch.cuda.current_stream().synchronize()
        tm['pass'] = time.perf_counter() - t_pass
        return res, tm
    # the samples are drawn on ONE host thread, chunk after chunk in problem order (the global numpy stream is consumed
    # exactly like the one-by-one loop), at most `workers + 1` chunks ahead of the device passes
    def _run_passes():
        nonlocal smooth_share
        with ThreadPoolExecutor(max_workers=1) as sampler, \
                ThreadPoolExecutor(max_workers=workers, initializer=bind_stream if workers > 1 else None) as pool:
            sampled, passes, submitted = {}, {}, set()

            def submit(ci):
                if ci < len(starts) and ci not in submitted:
                    submitted.add(ci)
                    sampled[ci] = sampler.submit(prepare, starts[ci])
                    if workers > 1:
                        passes[ci] = pool.submit(device_pass, ci, sampled[ci])
            for ci in range(min(workers + 1, len(starts))):
                submit(ci)
            for ci in range(len(starts)):
                if workers > 1:
                    res, tm = passes.pop(ci).result()
                else:
                    submit(ci + 1)
                    res, tm = device_pass(ci, sampled[ci])
                sampled.pop(ci, None)
                submit(ci + workers + 1)
                # smoothing share of THIS pass's own wall clock, weighted by its problem count (passes overlap in time under
                # workers > 1: summing their smoothing intervals would count the same wall-clock seconds twice)
                smooth_share += len(res) * min(tm.get('smoothing', 0.) / max(tm.get('pass', 0.), 1e-12), 1.0)
                _collect(res, 0., 0., sol, paths, smooth_paths, rows_out)
    t_begin = time.perf_counter()
    smooth_share = 0.
    starts = _pass_spans(len(indexes), chunk)
    import sys
    switch = sys.getswitchinterval()
    if workers > 1:
        # a worker coming back from a blocking device copy must not wait 5 ms (the default) for the interpreter lock while
        # another one runs a Python loop: a pass has dozens of such points
        sys.setswitchinterval(2e-4)
    try:
        _run_passes()
    finally:
        sys.setswitchinterval(switch)
    # wall clock of the whole evaluation (sampling of chunk i+1 overlaps the device pass of chunk i), spread evenly
    wall = (time.perf_counter() - t_begin) / max(len(sol), 1)
    # total_time_explore of eval_gnn.py: the per-problem wall clock minus the smoothing stage's share of it (a fraction in [0, 1])
    t_explore = wall * (1.0 - smooth_share / max(len(sol), 1))
    sol = [x[:5] + (wall, t_explore) for x in sol]
    n_success = sum(s[0] for s in sol)
    collision_explore = float(np.mean([s[3] for s in sol]))
    collision = float(np.mean([s[3] + s[4] for s in sol]))
    running_time = float(sum(s[5] for s in sol if s[0])) / max(n_success, 1)
    solution_cost = float(sum(s[2] for s in sol if s[0])) / max(n_success, 1)
    return (n_success, collision, running_time, solution_cost, sum(s[5] for s in sol), paths, smooth_paths,
            collision_explore, sum(s[6] for s in sol))


# --------------------------------------------------------------------------------------------------
# batched explore stage with everything but the sampling on the device (2-D mazes)
# --------------------------------------------------------------------------------------------------
def sample_maze_problems(problems, batch, k):
    """Host part of :func:`explore_maze_batch`: the reference's rejection sampling for every problem in order
    (``explore``: eval_gnn.py:180-184), consuming the global numpy RNG exactly like the one-by-one loop.  Returns
    (envs, node rows per problem [free incl. init / goal; collided], n_free, k1)."""
    from .graph_build import k1_of
    from .maze2d import AttemptStream, Maze2D
    envs, vs, n_free, k1s = [], [], [], []
    stream = AttemptStream()
    for pr in problems:
        env = Maze2D(np.asarray(pr['map'])[None], np.asarray(pr['init_state'])[None], np.asarray(pr['goal_state'])[None])
        env.init_new_problem(0)
        free, coll = env.sample_n_points_stream(stream, batch)                # same stream as sample_n_points
        coll = coll[:len(free)]                                               # eval_gnn.py:182 (before init / goal join)
        nf = len(free) + 2
        vrows = np.concatenate((np.asarray(env.init_state, dtype=np.float64).reshape(1, 2),
                                np.asarray(env.goal_state, dtype=np.float64).reshape(1, 2), free, coll)).astype(np.float32)
        envs.append(env)
        vs.append(torch.from_numpy(vrows))
        n_free.append(nf)
        k1s.append(k1_of(k, nf))
    stream.close()                                                            # global RNG: as if sampled one by one
    return envs, vs, n_free, k1s



_DRAWS_PER_FREE = [3.0]          # running estimate of uniform draws per free sample (sizes the block handed to the device)


def sample_maze_problems_device(problems, batch, k, device):
    """:func:`sample_maze_problems` with everything behind the random draws on the device (``gnnmp_maze_sample``,
    csrc/maze_kernels.hip): the host only draws the uniform stream (numpy's global generator, one block for the whole list of
    problems -- same values in the same order as the reference's one-by-one ``uniform_sample`` calls) and hands it over; the
    device classifies every draw, finds each problem's ``batch``-th free draw in stream order and writes the float32 node
    rows [start, goal, free ..., rejected[:batch] ...] (eval_gnn.py:180-184) where the graph builder reads them.  The global
    generator is left where one-by-one sampling would have left it.  Returns a dict for ``explore_maze_batch(presampled=...)``:
    the node rows never visit the host on their way to the explorer."""
    import ctypes
    from . import _lib
    from .graph_build import k1_of
    from .maze2d import AttemptStream, Maze2D
    B = len(problems)
    dev = torch.device(device)
    envs = []
    for pr in problems:
        env = Maze2D(np.asarray(pr['map'])[None], np.asarray(pr['init_state'])[None], np.asarray(pr['goal_state'])[None])
        env.init_new_problem(0)
        envs.append(env)
    w = int(np.asarray(problems[0]['map']).shape[0])
    maps = torch.from_numpy(np.ascontiguousarray(np.asarray([np.asarray(pr['map'], dtype=np.float64) for pr in problems]))).to(dev)
    init64 = torch.from_numpy(np.ascontiguousarray(np.asarray([np.asarray(e.init_state, dtype=np.float64).reshape(2) for e in envs]))).to(dev)
    goal64 = torch.from_numpy(np.ascontiguousarray(np.asarray([np.asarray(e.goal_state, dtype=np.float64).reshape(2) for e in envs]))).to(dev)
    v = torch.empty(B * (2 + 2 * batch), 2, dtype=torch.float32, device=dev)
    node_ptr = torch.empty(B + 1, dtype=torch.int32, device=dev)
    used = torch.empty(B, dtype=torch.int32, device=dev)
    state = torch.zeros(2, dtype=torch.int64, device=dev)                  # [cursor, ok (int32 in the low half)]
    stream = AttemptStream()
    m = int(B * batch * _DRAWS_PER_FREE[0] * 1.25) + 2048
    with torch.cuda.device(dev):
        while True:
            att = torch.from_numpy(stream.peek(m)).to(dev)
            state.zero_()
            sb = _lib.MazeSampleBatch(B, w, int(batch), int(att.shape[0]), att.data_ptr(), maps.data_ptr(), init64.data_ptr(),
                                      goal64.data_ptr())
            _lib.check(_lib.lib().gnnmp_maze_sample(ctypes.byref(sb), state.data_ptr(), v.data_ptr(), node_ptr.data_ptr(),
                                                    used.data_ptr(), state.data_ptr() + 8, torch.cuda.current_stream().cuda_stream),
                       'gnnmp_maze_sample')
            cursor, ok = state.cpu().tolist()                                # the one wait of the sampling
            if ok & 0xffffffff:
                break
            m *= 2                                                           # the block was too short: nothing was consumed
    used_h = used.cpu().numpy()
    nptr = node_ptr.cpu().numpy().astype(np.int64)
    stream.consume(int(cursor))
    stream.close()                                                           # global RNG: as if sampled one by one
    _DRAWS_PER_FREE[0] = max(1.5, 0.5 * _DRAWS_PER_FREE[0] + 0.5 * float(cursor) / max(B * batch, 1))
    for e, u in zip(envs, used_h):
        e.collision_check_count += int(u)                                    # maze_env.py: one check per draw
    nf = int(batch) + 2
    return {'envs': envs, 'v': v[:int(nptr[-1])], 'node_ptr': node_ptr, 'node_ptr_host': nptr, 'n_free': [nf] * B,
            'k1s': [k1_of(k, nf)] * B, 'maps': maps, 'goal64': goal64}


def maze_explore_device(v, node_ptr, edge_ptr, n_free, ei, scores, maps, goal64, resume=None, want_prev=False):
    """``gnnmp_maze_explore_ex`` on device tensors: greedy best-edge expansion + grid collision checks of B problems
    (``v`` [sum N, dim] float32 with dim 2 (point robot) or 3 (stick robot), ``ei`` [2, sum E] int64 graph-local,
    ``scores`` [sum E], ``maps`` [B, w, w] float64, ``goal64`` [B, dim] float64).  ``resume``: per-problem list of dicts
    (explored, prev {node: parent}, pairs [[a, b], ...]) of earlier rounds (eval_gnn.py:235-247) or None for fresh trees.
    Returns host-side (success, n_explored, n_pairs, path_len, checks) lists and the compacted ``explored`` /
    ``explored_edges`` (+ per-problem int offsets; in resume mode only the pairs added by this round) / ``path`` arrays,
    plus the parent array [sum N] when ``want_prev``."""
    import ctypes
    from . import _lib
    device = v.device
    B, w, dim = int(maps.shape[0]), int(maps.shape[1]), int(v.shape[1])
    nf = torch.as_tensor(n_free, dtype=torch.int32).to(device)
    total_n, total_e = int(v.shape[0]), int(ei.shape[1])
    mb = _lib.MazeBatch(B, total_n, total_e, w, v.data_ptr(), node_ptr.data_ptr(), edge_ptr.data_ptr(), nf.data_ptr(),
                        ei.data_ptr(), scores.data_ptr(), maps.data_ptr(), goal64.data_ptr())
    need = ctypes.c_size_t()
    _lib.check(_lib.lib().gnnmp_maze_explore_workspace_bytes(ctypes.byref(mb), ctypes.byref(need)), 'maze ws')
    ws = torch.empty(need.value, dtype=torch.uint8, device=device)
    i32 = lambda n: torch.zeros(n, dtype=torch.int32, device=device)      # noqa: E731
    success, n_expl, expl, n_pairs, ee, plen, path = i32(B), i32(B), i32(total_n), i32(B), i32(2 * (2 * total_e + B)), \
        i32(B), i32(total_n)
    checks = torch.zeros(B, dtype=torch.int64, device=device)
    prev_out = i32(total_n) if want_prev else None
    nptr_h = node_ptr.cpu().tolist()
    rs, keep = None, None
    if resume is not None:
        ne0 = np.array([len(r['explored']) for r in resume], dtype=np.int32)
        ex0 = np.zeros(total_n, dtype=np.int32)
        pv0 = np.zeros(total_n, dtype=np.int32)
        pptr = np.zeros(B + 1, dtype=np.int32)
        for b, r in enumerate(resume):
            ex0[nptr_h[b]:nptr_h[b] + len(r['explored'])] = r['explored']
            for node, par in r['prev'].items():
                pv0[nptr_h[b] + node] = par
            pptr[b + 1] = pptr[b] + len(r['pairs'])
        pairs = np.concatenate([np.asarray(r['pairs'], dtype=np.int32).reshape(-1) for r in resume])
        np0 = np.array([len(r['pairs']) for r in resume], dtype=np.int32)
        keep = [torch.from_numpy(a).to(device) for a in (ne0, ex0, pv0, np0, pairs, pptr)]
        rs = _lib.MazeResume(*(t.data_ptr() for t in keep))
    with torch.cuda.device(device):
        st = torch.cuda.current_stream().cuda_stream
        _lib.check(_lib.lib().gnnmp_maze_explore_ex(ctypes.byref(mb), dim, ctypes.byref(rs) if rs is not None else None,
                                                    success.data_ptr(), n_expl.data_ptr(), expl.data_ptr(), n_pairs.data_ptr(),
                                                    ee.data_ptr(), plen.data_ptr(), path.data_ptr(), checks.data_ptr(),
                                                    prev_out.data_ptr() if want_prev else None, ws.data_ptr(), ws.numel(), st),
                   'gnnmp_maze_explore_ex')
    success, n_expl, n_pairs, plen, checks = (t.cpu().tolist() for t in (success, n_expl, n_pairs, plen, checks))
    eptr = edge_ptr.cpu().tolist()
    # the pair list has room for 2E + 1 pairs per problem but holds a few hundred: compact it on the device
    ee_off = np.zeros(B + 1, dtype=np.int64)
    ee_off[1:] = np.cumsum([2 * n for n in n_pairs])
    take = np.concatenate([np.arange(2 * (2 * eptr[b] + b), 2 * (2 * eptr[b] + b) + 2 * n_pairs[b], dtype=np.int64)
                           for b in range(B)])
    ee = ee[torch.from_numpy(take).to(device)].cpu().numpy() if take.size else np.zeros(0, dtype=np.int32)
    out = (success, n_expl, n_pairs, plen, checks, expl.cpu().numpy(), ee, ee_off, path.cpu().numpy())
    return out + (prev_out.cpu().numpy(),) if want_prev else out



@torch.no_grad()
def explore_maze_batch(problems, model, device, batch=500, k=30, loop=5, model_s=None, smooth_iters=5, timings=None,
                       presampled=None):
    """Many 2-D maze problems at once: sampling on the host (the reference's numpy RNG
    stream, one problem after the other), then -- in ONE pass on the device -- kNN graphs
    (graph_kernels.hip), explorer forward (batched), greedy expansion + collision checks
    (maze_kernels.hip) and, when ``model_s`` is given, the smoothing stage of the solved problems
    (smoother.py:233-246: ``smooth_iters`` x (batched smoother forward, collision-checked steering on
    the device)).  ``problems``: list of dicts(map [w, w], init_state, goal_state).
    Covers the reference's default single-forward case (batch == t_max); returns one
    result dict per problem with the fields of ``explore`` (``smooth_path`` / ``c_smooth`` with ``model_s``).
    ``timings``: optional dict that receives wall-clock seconds per stage (adds device syncs).
    ``presampled``: the result of :func:`sample_maze_problems` for these problems (lets a caller sample the next
    chunk on the host while the device works on this one)."""
    def mark(name, t_prev):
        if timings is None:
            return t_prev
        torch.cuda.current_stream().synchronize()      # (this pass's stream only: other passes may be in flight)
        now = time.perf_counter()
        timings[name] = timings.get(name, 0.) + now - t_prev
        return now
    tm = time.perf_counter()
    from .batch import GraphBatch
    from .graph_build import build_edges_gpu
    B = len(problems)
    dsamp = presampled if isinstance(presampled, dict) else None          # sample_maze_problems_device: the rows are on the device
    if dsamp is not None:
        envs, n_free, k1s = dsamp['envs'], dsamp['n_free'], dsamp['k1s']
        v, node_ptr = dsamp['v'], dsamp['node_ptr']
        ptr = torch.from_numpy(np.asarray(dsamp['node_ptr_host'], dtype=np.int64))
        vs = None
        tm = mark('host_sampling', tm)
    else:
        envs, vs, n_free, k1s = presampled if presampled is not None else sample_maze_problems(problems, batch, k)
        tm = mark('host_sampling', tm)
        ptr = torch.zeros(B + 1, dtype=torch.int64)
        ptr[1:] = torch.tensor([x.shape[0] for x in vs]).cumsum(0)
        node_ptr = ptr.to(torch.int32).to(device)
        v = torch.cat(vs).to(device)
    ei, edge_ptr = build_edges_gpu(v, node_ptr, n_free, k1s)
    tm = mark('graph_build', tm)
    obs = [np.asarray(e.obstacles).reshape(-1, 2) for e in envs]
    ocount = np.array([o.shape[0] for o in obs], dtype=np.int64)
    optr = np.zeros(B + 1, dtype=np.int32)
    optr[1:] = np.cumsum(ocount)
    goals = torch.tensor(np.asarray([e.goal_state for e in envs]), dtype=torch.float32).to(device)
    # (the per-problem arrays come from np.argwhere and are column-major: the joined array is made row-major explicitly)
    obs_all = np.ascontiguousarray(np.concatenate(obs), dtype=np.float32)
    gb = GraphBatch(v, goals, torch.from_numpy(obs_all).to(device), ei, node_ptr, edge_ptr,
                    torch.from_numpy(optr).to(device), int(ocount.max()))
    scores = model.forward_batch(gb, loop)
    tm = mark('explorer_forward', tm)
    w = int(np.asarray(problems[0]['map']).shape[0])
    if dsamp is not None:
        maps, goal64 = dsamp['maps'], dsamp['goal64']
    else:
        maps = torch.tensor(np.asarray([np.asarray(pr['map'], dtype=np.float64) for pr in problems])).to(device)
        goal64 = torch.tensor(np.asarray([e.goal_state for e in envs], dtype=np.float64)).to(device)
    success, n_expl, n_pairs, plen, checks, expl, ee, ee_off, path = maze_explore_device(
        v, node_ptr, edge_ptr, n_free, ei, scores, maps, goal64)
    nptr = ptr.tolist()
    tm = mark('greedy_explore', tm)
    smoothed = {}
    if model_s is not None and any(success):
        smoothed = _smooth_maze_batch(model_s, [b for b in range(B) if success[b]], v, nptr, n_free, path, plen, maps, w,
                                      smooth_iters, device)
        tm = mark('smoothing', tm)
    out = []
    if vs is None:                                                       # device sampling: one copy of the node rows for the results
        v_host = v.cpu().numpy()
        vs_np = [v_host[nptr[b]:nptr[b + 1]] for b in range(B)]
        vs = [torch.from_numpy(x) for x in vs_np]
    else:
        vs_np = [x.numpy() for x in vs]
    for b in range(B):
        nodes = path[nptr[b]:nptr[b] + plen[b]]
        # results stay numpy arrays (explored [n], explored_edges [m, 2], path [P, 2]): building Python lists for
        # hundreds of problems costs more than the device pass itself
        out.append({'success': bool(success[b]), 'explored': expl[nptr[b]:nptr[b] + n_expl[b]],
                    'explored_edges': ee[ee_off[b]:ee_off[b + 1]].reshape(-1, 2),
                    'c_explore': envs[b].collision_check_count + int(checks[b]),
                    'path': vs_np[b][nodes], 'free': None, 'env': envs[b], 'v': vs[b], 'n_free': n_free[b]})
        if model_s is not None:
            sp, cs = smoothed.get(b, (np.zeros((0, 2), dtype=np.float32), 0))
            out[-1].update(smooth_path=sp, c_smooth=cs)
    mark('results', tm)
    return out


def _smooth_maze_batch(model_s, sel, v, nptr, n_free, path, plen, maps, w, iters, device):
    """Smoothing stage of the solved problems ``sel`` entirely on the device: per iteration one batched
    smoother forward (loop = 1) and one ``gnnmp_maze_steer`` launch; one D2H copy at the end.
    Samples handed to the network: the first 500 free and 500 collided points (smoother.py:52-64)."""
    from . import _lib
    from .smoother import SmoothBatch
    total_n = int(v.shape[0])
    v_ext = torch.cat((v, torch.zeros(1, 2, device=device)))          # row total_n: the reference's zero filler row
    widx, fidx, cidx, pc, fc, cc, ec = [], [], [], [], [], [], []
    for b in sel:
        nb = nptr[b + 1] - nptr[b]
        widx.append(path[nptr[b]:nptr[b] + plen[b]].astype(np.int64) + nptr[b])
        nf, nc = min(n_free[b], 500), min(nb - n_free[b], 500)
        fidx.append(np.arange(nf, dtype=np.int64) + nptr[b] if nf else np.array([total_n], dtype=np.int64))
        cidx.append(np.arange(nc, dtype=np.int64) + nptr[b] + n_free[b] if nc else np.array([total_n], dtype=np.int64))
        pc.append(plen[b]); fc.append(len(fidx[-1])); cc.append(len(cidx[-1])); ec.append(3 * plen[b] - 2)
    take = lambda parts: v_ext[torch.from_numpy(np.concatenate(parts)).to(device)]      # noqa: E731
    sb = SmoothBatch.from_device(take(widx), take(fidx), take(cidx), torch.from_numpy(_chain_edge_indices(pc)).to(device),
                                 pc, fc, cc, ec)
    maps_sel = maps[torch.tensor(sel, device=device)].contiguous()
    checks = torch.zeros(len(sel), dtype=torch.int64, device=device)
    tmp = torch.empty_like(sb.path)
    L = _lib.lib()
    with torch.cuda.device(device):
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(iters):
            new = model_s.forward_batch(sb, 1)
            out = torch.empty_like(sb.path)
            _lib.check(L.gnnmp_maze_steer(len(sel), int(sb.path.shape[0]), w, maps_sel.data_ptr(), sb.path_ptr.data_ptr(),
                                          sb.path.data_ptr(), new.data_ptr(), out.data_ptr(), tmp.data_ptr(),
                                          checks.data_ptr(), st), 'gnnmp_maze_steer')
            sb.path = out
    final, checks = sb.path.cpu().numpy(), checks.cpu().tolist()
    res, o = {}, 0
    for i, b in enumerate(sel):
        res[b] = (final[o:o + plen[b]], int(checks[i]))
        o += plen[b]
    return res


# --------------------------------------------------------------------------------------------------
# the general planner loop on the device: resample rounds (t_max > batch) and the 3-DoF maze
# --------------------------------------------------------------------------------------------------
def _device_round(states, model, device, k, loop, fresh):
    """One explorer forward + one greedy expansion round for a list of per-problem states (see
    :func:`eval_gnn_device_rounds`): graphs over the CURRENT sample sets, batched forward, device explore with the
    trees of earlier rounds carried over unless ``fresh``.  Updates the states in place."""
    from .batch import GraphBatch
    from .graph_build import build_edges_gpu, k1_of
    dim = states[0]['env'].config_dim
    vs, n_free, k1s = [], [], []
    for st in states:
        st['v'] = np.concatenate((np.asarray(st['free'], dtype=np.float64).reshape(-1, dim),
                                  np.asarray(st['coll'], dtype=np.float64).reshape(-1, dim))).astype(np.float32)
        vs.append(torch.from_numpy(st['v']))
        n_free.append(len(st['free']))
        k1s.append(k1_of(k, len(st['free'])))
    B = len(states)
    ptr = torch.zeros(B + 1, dtype=torch.int64)
    ptr[1:] = torch.tensor([x.shape[0] for x in vs]).cumsum(0)
    node_ptr = ptr.to(torch.int32).to(device)
    v = torch.cat(vs).to(device)
    ei, edge_ptr = build_edges_gpu(v, node_ptr, n_free, k1s)
    obs = [torch.tensor(np.asarray(st['env'].obstacles), dtype=torch.float32).reshape(-1, 2) for st in states]
    optr = torch.zeros(B + 1, dtype=torch.int64)
    optr[1:] = torch.tensor([o.shape[0] for o in obs]).cumsum(0)
    goals = torch.tensor(np.asarray([st['env'].goal_state for st in states]), dtype=torch.float32).to(device)
    gb = GraphBatch(v, goals, torch.cat(obs).to(device), ei, node_ptr, edge_ptr, optr.to(torch.int32).to(device),
                    max(o.shape[0] for o in obs))
    scores = model.forward_batch(gb, loop)
    maps = torch.tensor(np.asarray([np.asarray(st['env'].map, dtype=np.float64) for st in states])).to(device)
    goal64 = torch.tensor(np.asarray([st['env'].goal_state for st in states], dtype=np.float64)).to(device)
    resume = None if fresh else [{'explored': st['explored'], 'prev': st['prev'], 'pairs': st['pairs']} for st in states]
    success, n_expl, n_pairs, plen, checks, expl, ee, ee_off, path, prev = maze_explore_device(
        v, node_ptr, edge_ptr, n_free, ei, scores, maps, goal64, resume=resume, want_prev=True)
    nptr = ptr.tolist()
    for b, st in enumerate(states):
        st['explored'] = expl[nptr[b]:nptr[b] + n_expl[b]].tolist()
        new_pairs = ee[ee_off[b]:ee_off[b + 1]].reshape(-1, 2).tolist()
        st['pairs'] = new_pairs if fresh else st['pairs'] + new_pairs
        st['prev'] = {a: int(prev[nptr[b] + a]) for a in st['explored']}
        st['checks'] += int(checks[b])
        st['success'] = bool(success[b])
        st['path_nodes'] = path[nptr[b]:nptr[b] + plen[b]].tolist()
        st['rounds'] += 1


def eval_gnn_device_rounds(env, indexes, model, model_s=None, seed=1234, batch=500, t_max=500, k=30, device='cuda', loop=5,
                           chunk=64, rows_out=None):
    """:func:`eval_gnn` (eval_gnn.py:96-145) with the planner on the device for the GENERAL loop of ``explore``
    (eval_gnn.py:191-247): when the frontier dies the problem gets ``batch`` more samples, the explorer runs again on
    the larger graph and the search tree carries over -- up to ``t_max`` free samples -- for ``Maze2D`` and ``Maze3D``
    environments (``model_s`` = None: the reference's smoother='none' branch, as maze3 has no shipped smoother).

    The reference draws every sample from ONE global numpy stream, problem after problem, and a problem that needs a
    second round draws it before the next problem's first.  Batched rounds keep that order by speculation: a chunk of
    problems is sampled and explored as if nobody needed a second round; the first problem that does is rewound to
    the stream position right after ITS first sampling, finished on its own (device rounds with the tree carried over),
    and the problems behind it -- whose samples came from the wrong stream position -- are sampled and explored again
    from the position the sequential loop would have reached.  Per-problem outcomes therefore equal the one-by-one loop's
    (tests/test_planner_rounds_gpu.py against rows recorded from the unmodified reference)."""
    model.eval()                       # eval_gnn.py:109-110
    if model_s is not None:
        model_s.eval()
    np.random.seed(seed)
    torch.manual_seed(seed)
    dim = env.config_dim
    results = {}

    def sample(e, n):
        if dim == 2:
            f, c = e.sample_n_points_arrays(n)
            return list(f), list(c)
        return e.sample_n_points(n, need_negative=True)

    def new_state(i):
        e = type(env)(np.asarray(env.maps[i])[None], np.asarray(env.init_states[i])[None], np.asarray(env.goal_states[i])[None])
        e.init_new_problem(0)
        free, coll = sample(e, batch)
        coll = coll[:len(free)]                                         # eval_gnn.py:180 (before init / goal join)
        free = [np.asarray(e.init_state, dtype=np.float64), np.asarray(e.goal_state, dtype=np.float64)] + free
        return {'i': i, 'env': e, 'free': free, 'coll': coll, 'explored': [0], 'prev': {0: 0}, 'pairs': [[0, 0]],
                'checks': 0, 'success': False, 'rounds': 0, 'path_nodes': []}

    def may_resample(st):
        return (batch + len(st['free']) - 2) <= t_max                   # eval_gnn.py:239-240

    todo = list(indexes)
    while todo:
        part = todo[:chunk]
        states, rng_after = [], []
        for i in part:
            states.append(new_state(i))
            rng_after.append(np.random.get_state())
        _device_round(states, model, device, k, loop, fresh=True)
        done = len(part)
        for pos, st in enumerate(states):
            if st['success'] or not may_resample(st):
                results[st['i']] = st
                continue
            np.random.set_state(rng_after[pos])                          # where the sequential loop stands now
            while not st['success'] and may_resample(st):
                nf, nc = sample(st['env'], batch)
                st['free'] = st['free'] + nf
                st['coll'] = (st['coll'] + nc)[:len(st['free'])]        # eval_gnn.py:243-245
                _device_round([st], model, device, k, loop, fresh=False)
            results[st['i']] = st
            done = pos + 1                                               # samples of the problems behind are stale
            break
        todo = todo[done:]
    out = [results[i] for i in indexes]
    smoothed = {}
    if model_s is not None and dim == 2:
        sel = [b for b, st in enumerate(out) if st['success']]
        if sel:
            vcat = torch.cat([torch.from_numpy(st['v']) for st in out]).to(device)
            nptr = np.concatenate(([0], np.cumsum([st['v'].shape[0] for st in out]))).tolist()
            path = np.zeros(nptr[-1], dtype=np.int32)
            for b, st in enumerate(out):
                path[nptr[b]:nptr[b] + len(st['path_nodes'])] = st['path_nodes']
            maps = torch.tensor(np.asarray([np.asarray(st['env'].map, dtype=np.float64) for st in out])).to(device)
            smoothed = _smooth_maze_batch(model_s, sel, vcat, nptr, [len(st['free']) for st in out], path,
                                          [len(st['path_nodes']) for st in out], maps, int(maps.shape[1]), 5, device)
    sol = []
    for b, st in enumerate(out):
        p = st['v'][st['path_nodes']] if st['success'] else np.zeros((0, dim), dtype=np.float32)
        sp, cs = smoothed.get(b, (p, 0)) if st['success'] else (p, 0)
        c_explore = st['env'].collision_check_count + st['checks']
        sol.append((int(st['success']), path_cost(p), path_cost(sp), c_explore, cs, len(p), len(st['explored']), st['rounds']))
        if rows_out is not None:
            rows_out.append(sol[-1][:7])
    n_success = sum(s[0] for s in sol)
    return {'n_success': n_success, 'collision_explore': float(np.mean([s[3] for s in sol])),
            'collision': float(np.mean([s[3] + s[4] for s in sol])),
            'solution_cost': float(sum(s[2] for s in sol if s[0])) / max(n_success, 1), 'rounds': [s[7] for s in sol]}

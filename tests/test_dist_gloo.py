"""Multi-process path (world_size 2, gloo, CPU): problem sharding has no data-path collective, so
correctness = every shard computes exactly what it would compute alone, and the final result gather
returns the concatenation in rank order."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import gnnmp  # noqa: F401
from gnnmp.dist import gather_problem_results, gather_variable, shard_range


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _fake_scores(lo, hi):
    """Deterministic stand-in for per-problem edge scores (problem i has 5 + i % 3 edges)."""
    out = []
    for i in range(lo, hi):
        g = torch.Generator().manual_seed(i)
        out.append(torch.rand(5 + i % 3, generator=g))
    return torch.cat(out) if out else torch.zeros(0)


def _worker(rank, world, port, n_problems, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    lo, hi = shard_range(n_problems, rank, world)
    local = _fake_scores(lo, hi)
    parts = gather_variable(local)
    rows = torch.tensor([[float(i), float(i * i)] for i in range(lo, hi)], dtype=torch.float64).reshape(-1, 2)
    allrows = gather_problem_results(rows)
    q.put((rank, lo, hi, [p.numpy().copy() for p in parts], allrows.numpy().copy()))      # numpy: no fd passing of tensors
    dist.barrier()
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize('world,n', [(2, 7), (3, 2)])       # (3, 2): one rank owns nothing -- an empty shard in the padded gather
def test_two_rank_gather_equals_single_process(world, n):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    whole = _fake_scores(0, n)
    for rank, lo, hi, parts, allrows in got:
        parts = [torch.from_numpy(p) for p in parts]
        allrows = torch.from_numpy(allrows)
        assert torch.equal(torch.cat(parts), whole)                 # same bytes as the unsharded run
        assert torch.equal(parts[rank], _fake_scores(lo, hi))       # a shard run alone gives its slice
        assert torch.equal(allrows[:, 0], torch.arange(n, dtype=torch.float64))
        assert torch.equal(allrows[:, 1], torch.arange(n, dtype=torch.float64) ** 2)


def _rows_worker(rank, world, port, fixture, q):
    import numpy as np
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    with np.load(fixture) as f:
        rows = f['rows']
    lo, hi = shard_range(rows.shape[0], rank, world)          # what eval_gnn_device(shard=(rank, world)) evaluates
    allrows = gather_problem_results(torch.from_numpy(rows[lo:hi].copy()))
    q.put((rank, allrows.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_gather_recorded_device_planner_rows():
    """Real per-problem rows (success, path cost, smoothed cost, c_explore, c_smooth, path length, explored:
    eval_gnn.py:120-122) recorded from planner.eval_gnn_device on an MI355X (tools/record_device_rows.py); rank r owns
    the contiguous block eval_gnn_device(shard=(r, 2)) evaluates (the GPU test tests/test_dist_gpu.py runs exactly that
    with two processes and compares with this fixture); the gather restores the sequential order and the aggregates
    of eval_gnn.py:128-145."""
    import numpy as np
    fixture = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'evalrows_device_first64.npz')
    with np.load(fixture) as f:
        want = f['rows']
    assert want.shape == (64, 7)
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rows_worker, args=(r, world, port, fixture, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, allrows in got:
        assert np.array_equal(allrows, want)
        assert int(allrows[:, 0].sum()) == int(want[:, 0].sum())
        assert float(np.mean(allrows[:, 3] + allrows[:, 4])) == float(np.mean(want[:, 3] + want[:, 4]))


def test_shard_ranges_cover_and_balance():
    for n in (0, 1, 7, 256, 1000):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    # weighted split: cost-balanced by edge count
    w = [1] * 50 + [10] * 50
    spans = [shard_range(100, r, 2, weights=w) for r in range(2)]
    assert spans[0][1] == spans[1][0] and spans[1][1] == 100
    cost = [sum(w[a:b]) for a, b in spans]
    assert abs(cost[0] - cost[1]) <= 10


def test_gather_without_process_group():
    x = torch.arange(5.)
    assert torch.equal(gather_variable(x)[0], x)

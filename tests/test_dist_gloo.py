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


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE configs[3]: family-aware sharding of a mixed problem set (gnnmp.dist.mixed_plan / shard_mixed)
# ---------------------------------------------------------------------------------------------------------------------
_FAM = {'maze2': (2, 32, 2, 116, 11270), 'snake7': (7, 32, 2, 116, 12321), 'ur5': (6, 32, 6, 5, 12101), 'kuka7': (7, 64, 6, 5, 12291)}


def _mixed_job(n=256, jitter=True):
    """families interleaved the way a mixed evaluation set arrives; per-problem edge counts jittered around the measured means"""
    from gnnmp.dist import forward_cost
    names = list(_FAM)
    fams = [names[i % 4] for i in range(n)]
    g = torch.Generator().manual_seed(7)
    costs = []
    for f in fams:
        C, d, S, O, E = _FAM[f]
        e = int(E * (1 + (0.1 * (torch.rand(1, generator=g).item() - 0.5) if jitter else 0)))
        costs.append(forward_cost(1000, e, O, C, d, S))
    return fams, costs


def _fixed():
    from gnnmp.dist import batch_time
    return {f: (lambda g, d=_FAM[f][1]: batch_time(d, 'fp32', g)) for f in _FAM}


def test_mixed_plan_is_a_partition_with_few_families_per_rank():
    from gnnmp.dist import mixed_plan, plan_times
    fams, costs = _mixed_job()
    fixed = _fixed()
    for world in (1, 2, 3, 4, 8):
        plan = mixed_plan(fams, costs, world, fixed)
        assert len(plan) == world
        assert sorted(i for r in plan for i in r) == list(range(len(fams)))               # a partition of the job
        loads = plan_times(plan, fams, costs, fixed)                                      # problems + one fixed term per family held
        assert max(loads) <= 1.10 * sum(loads) / world, (world, loads)                   # predicted-time imbalance <= 10 %
        if world >= 4:
            assert max(len({fams[i] for i in r}) for r in plan) <= 2, world               # <= 2 kernel instantiations per rank
        for r in plan:                                                                    # caller order is kept inside a family
            for f in set(fams[i] for i in r):
                mine = [i for i in r if fams[i] == f]
                assert mine == sorted(mine)
    # time share, not edge share: at equal edge counts a d = 64 kuka7 problem takes ~2.4 x the time of a d = 32 one, so the ranks
    # that hold kuka7 hold FEWER problems
    plan = mixed_plan(fams, costs, 8, fixed)
    kuka_only = [r for r in plan if {fams[i] for i in r} == {'kuka7'}]
    d32_only = [r for r in plan if r and 'kuka7' not in {fams[i] for i in r}]
    assert kuka_only and d32_only and max(map(len, kuka_only)) < min(map(len, d32_only))


def test_mixed_plan_does_not_slice_a_small_family_thinner_than_its_fixed_cost():
    """The fixed part of a family batch (launch chain, under-filled tails: dist.batch_time does not go through the origin) is paid
    once per rank that holds the family.  A job whose kuka7 share is 8 problems is cheaper with those 8 on ONE rank than cut into slivers that each pay
    the fixed term again -- the round-5 model (FLOPs x a constant, no fixed term) sliced them (measured slowest / mean 1.263 where it
    predicted 1.018, profiles/r05_cfg4_mixed.txt)."""
    from gnnmp.dist import forward_cost, mixed_plan, plan_times
    fixed = _fixed()
    fams = ['kuka7'] * 8 + ['maze2'] * 248
    costs = [forward_cost(1000, _FAM[f][4], _FAM[f][3], _FAM[f][0], _FAM[f][1], _FAM[f][2]) for f in fams]
    plan = mixed_plan(fams, costs, 8, fixed)
    holders = [r for r in plan if any(fams[i] == 'kuka7' for i in r)]
    assert len(holders) == 1                                                             # not sliced
    t = plan_times(plan, fams, costs, fixed)
    assert max(t) <= 1.10 * sum(t) / 8
    # without time curves the same call is the plain cost-share split (nobody more than one problem above the mean share)
    plan0 = mixed_plan(fams, costs, 8)
    t0 = plan_times(plan0, fams, costs)
    assert max(t0) <= sum(t0) / 8 + max(costs)
    # fewer problems than ranks: nobody gets more than one, the rest idle
    plan1 = mixed_plan(fams[:5], costs[:5], 8, fixed)
    assert sorted(len(r) for r in plan1) == [0, 0, 0, 1, 1, 1, 1, 1]


def _mixed_worker(rank, world, port, q):
    from gnnmp.dist import mixed_plan
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    fams, costs = _mixed_job(256)
    plan = mixed_plan(fams, costs, world, _fixed())        # every rank derives the same plan from the same metadata
    mine = plan[rank]
    local = torch.cat([_fake_scores(i, i + 1) for i in mine]) if mine else torch.zeros(0)
    parts = gather_variable(local)
    # reassemble in CALLER order from the plan alone (no index exchange)
    out = [None] * len(fams)
    for r, part in enumerate(parts):
        off = 0
        for i in plan[r]:
            n = 5 + i % 3
            out[i] = part[off:off + n]
            off += n
    from gnnmp.dist import plan_times
    q.put((rank, [fams[i] for i in mine], plan_times([mine], fams, costs, _fixed())[0], torch.cat(out).numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_eight_ranks_family_aware_shards_and_gather():
    """world size 8 over gloo: every rank scores the problems mixed_plan gives it (stand-in scores), the padded gather returns
    every rank's block, and the job's scores come back in the caller's problem order; <= 2 families per rank, predicted
    time within 10 % of the mean."""
    world = 8
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mixed_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    whole = _fake_scores(0, 256)
    loads = {}
    for rank, fam_list, load, flat in got:
        assert torch.equal(torch.from_numpy(flat), whole)
        assert len(set(fam_list)) <= 2
        loads[rank] = load
    mean = sum(loads.values()) / world
    assert max(loads.values()) <= 1.10 * mean, loads


def _strong_worker(rank, world, port, n_total, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    lo, hi = shard_range(n_total, rank, world)             # bench.py --strong N_TOTAL: problem i = seed 1234 + i on whatever rank
    local = _fake_scores(1234 + lo, 1234 + hi)
    parts = gather_variable(local)
    ones = torch.ones(1, dtype=torch.float64)
    dist.all_reduce(ones)                                  # bench.py's `ranks_seen`
    q.put((rank, hi - lo, int(ones.item()), torch.cat(parts).numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_strong_split_scores_the_same_fixed_set(world):
    """bench.py --strong: the job is a FIXED problem set whatever the rank count -- the gathered scores of a 2- or 3-rank run are
    the bytes of the one-rank run, every rank holds its contiguous block, and the all_reduce of ones counts the ranks."""
    n_total = 10
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_strong_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    whole = _fake_scores(1234, 1234 + n_total)
    assert sum(g[1] for g in got) == n_total
    for rank, n_local, seen, flat in got:
        assert seen == world
        assert torch.equal(torch.from_numpy(flat), whole)

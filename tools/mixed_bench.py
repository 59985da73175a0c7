#!/usr/bin/env python
"""Run ON THE GPU BOX: BASELINE configs[3] (mixed maze / snake / ur5 / kuka set, 1000-node k1 = 8, fp32) on one GPU -- graphs/s of
the whole mixed job through gnnmp.dist.run_mixed and per family (64 problems each), kept as profiles/rNN_cfg4_mixed.txt."""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import gnnmp  # noqa: E402
from gnnmp.dist import run_mixed  # noqa: E402
from gnnmp.synth import ENVS, synth_batch_gpu  # noqa: E402
from gnnmp.weights import load_weights  # noqa: E402

DEV = 'cuda:0'
FAMILIES = ['maze2', 'snake7', 'ur5', 'kuka7']
PER, N, K1, LOOP = 64, 1000, 8, 5
dtype = sys.argv[1] if len(sys.argv) > 1 else 'fp32'
problems, models = [], {}
for fi, env in enumerate(FAMILIES):
    e = ENVS[env]
    m = gnnmp.EncoderProcessDecoder(e['workspace'], e['C'], e['d'], e['S']).eval()
    m.load_state_dict(load_weights(e['ckpt']), strict=True)
    m.mlp_dtype = dtype
    models[env] = m
    problems += [dict(env=env, **g) for g in synth_batch_gpu(env, N, K1, PER, DEV, seed0=5000 + 1000 * fi)]


def rate(ps, reps=10):
    for _ in range(3):
        run_mixed(ps, models, loop=LOOP)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        run_mixed(ps, models, loop=LOOP)
    torch.cuda.synchronize()
    return len(ps) * reps / (time.perf_counter() - t0)


print('BASELINE configs[3] on one MI355X, %s: %d problems, %d-node k1=%d RGGs, loop %d' % (dtype, len(problems), N, K1, LOOP))
print('%-10s %4s %4s %4s %8s %12s' % ('family', 'C', 'd', 'O', 'mean E', 'graphs/s'))
for env in FAMILIES:
    ps = [p for p in problems if p['env'] == env]
    e = ENVS[env]
    print('%-10s %4d %4d %4d %8.0f %12.1f' % (env, e['C'], e['d'], ps[0]['obstacles'].reshape(-1, e['S']).shape[0],
                                            sum(p['edge_index'].shape[1] for p in ps) / len(ps), rate(ps)))
print('%-10s %35s %12.1f   (run_mixed: GraphBatch assembly + one batched forward per family, one stream)' % ('mixed job', '', rate(problems)))


def job_rate(job, reps=20):
    for _ in range(3):
        job.run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            job.run()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return len(problems) * reps / sorted(ts)[1]


from gnnmp.dist import MixedJob, family_time_curves, mixed_plan, plan_times, problem_costs  # noqa: E402
ref = run_mixed(problems, models, loop=LOOP)
for conc in (False, True):
    job = MixedJob(problems, models, loop=LOOP, concurrent=conc)
    same = all(torch.equal(a, b) for a, b in zip(job.run(), ref))
    print('%-10s %35s %12.1f   (MixedJob: batches resident, %s; bytes equal run_mixed: %s)' % (
        'mixed job', '', job_rate(job), 'one stream per family, most expensive first' if conc else 'families back to back on one stream', same))
costs = problem_costs(problems, models, LOOP)
fixed = family_time_curves(models)
fams = [p['env'] for p in problems]
fam_cost = {env: sum(c for c, p in zip(costs, problems) if p['env'] == env) / PER for env in FAMILIES}
print('cost model (gnnmp.dist.batch_time): predicted ms of a 16 / 64-problem batch: ' + ', '.join('%s %.3f / %.3f' % (e, fixed[e](16 * fam_cost[e]), fixed[e](64 * fam_cost[e])) for e in FAMILIES))
for world in (2, 4, 8):
    plan = mixed_plan(fams, costs, world, fixed)
    loads = plan_times(plan, fams, costs, fixed)
    print('shard_mixed over %d ranks: problems per rank %s, families per rank %s, predicted slowest / mean %.3f' % (
        world, [len(r) for r in plan], [len({problems[i]['env'] for i in r}) for r in plan], max(loads) / (sum(loads) / world)))
    if world == 8:
        # what ONE rank of the 8-rank job runs, measured on this GPU: the slowest shard bounds the job
        rates = []
        for r in plan:
            j = MixedJob([problems[i] for i in r], models, loop=LOOP)
            for _ in range(5):
                j.run()
            ts = []
            for _ in range(5):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(20):
                    j.run()
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) / 20 * 1e3)
            rates.append(sorted(ts)[2])                      # median of five blocks of 20 runs
        print('   predicted ms per shard: %s' % ['%.2f' % x for x in loads])
        print('   measured ms per shard on this GPU: %s -> slowest / mean %.3f; whole job on one GPU / slowest shard = strong scaling %.2fx on 8' % (
            ['%.2f' % x for x in rates], max(rates) / (sum(rates) / len(rates)), (len(problems) / job_rate(MixedJob(problems, models, loop=LOOP)) * 1e3) / max(rates)))

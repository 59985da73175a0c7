"""BASELINE configs[3] at FULL size on one GPU: the mixed environment set (maze / snake / ur5 / kuka, constructor sites
str2name.py:14,22,38,46 of the reference) -- 256 problems, 64 per family, 1000-node k1 = 8 RGGs, fp32 -- scored through
``gnnmp.dist.run_mixed`` (one batched forward per family, results in the caller's order).  Size-independent properties:
(a) two runs give identical bytes; (b) a graph scored inside the mixed job equals the same graph scored alone, bit for bit,
for every family; (c) one sampled graph per family agrees with the CPU oracle within the absolute fp32 bar of
tests/parity_bar.py; (d) the resident, stream-per-family form (gnnmp.dist.MixedJob) gives the bytes of run_mixed; (e) the family-aware shards of the job
(gnnmp.dist.shard_mixed: bucketed by family, cut at cost quantiles) partition it, hold at most two families per rank, balance the
predicted time within 10 % and score their problems to the same bytes as the whole job does.  kuka7 (d = 64) runs the fp32 d = 64 kernels here -- a BASELINE config member."""
import pytest
import torch

from conftest import load_weights
import gnnmp
from gnnmp.dist import MixedJob, family_time_curves, plan_times, problem_costs, run_mixed, shard_mixed, shard_range
from gnnmp.synth import ENVS, synth_batch_gpu
from parity_bar import assert_fp32_parity, explorer_oracle_pair

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
FAMILIES = ['maze2', 'snake7', 'ur5', 'kuka7']
PER_FAMILY, N, K1, LOOP = 64, 1000, 8, 5


def _job():
    problems = []
    for fi, env in enumerate(FAMILIES):
        for g in synth_batch_gpu(env, N, K1, PER_FAMILY, DEV, seed0=5000 + 1000 * fi):
            problems.append(dict(env=env, **g))
    # interleave the families the way a mixed evaluation set arrives
    order = [f * PER_FAMILY + i for i in range(PER_FAMILY) for f in range(len(FAMILIES))]
    problems = [problems[i] for i in order]
    models = {}
    for env in FAMILIES:
        e = ENVS[env]
        m = gnnmp.EncoderProcessDecoder(e['workspace'], e['C'], e['d'], e['S']).eval()
        m.load_state_dict(load_weights(e['ckpt']), strict=True)
        models[env] = m
    return problems, models


def test_cfg4_mixed_set_full_size():
    problems, models = _job()
    assert len(problems) == 256
    s1 = [s.clone() for s in run_mixed(problems, models, loop=LOOP)]
    s2 = run_mixed(problems, models, loop=LOOP)
    assert all(torch.equal(a, b) for a, b in zip(s1, s2))                               # (a)
    for fi, env in enumerate(FAMILIES):
        idx = 4 * 17 + fi                                                               # the 18th problem of each family
        p = problems[idx]
        assert p['env'] == env
        alone = models[env].edge_scores(p['goal'], LOOP, p['v'], p['obstacles'], p['edge_index'])
        assert torch.equal(alone, s1[idx]), env                                         # (b)
        g = {k: p[k].cpu() for k in ('v', 'goal', 'obstacles', 'edge_index')}
        ref32, ref64 = explorer_oracle_pair(load_weights(ENVS[env]['ckpt']), g, LOOP)
        c = assert_fp32_parity(s1[idx].cpu(), ref32, ref64, env)                        # (c)
        print('%s N=%d E=%d: |gpu-ref64| %.2e  |gpu-ref32| %.2e  oracle fp32-vs-fp64 %.2e  bar %.2e' % (
            env, N, g['edge_index'].shape[1], c['err64'], c['err32'], c['own'], c['atol']))
    # (d) the resident form of the job -- batches built once, every family's forward on its own stream (gnnmp.dist.MixedJob) --
    # gives the bytes of run_mixed, with and without the concurrent streams, run after run
    for concurrent in (True, False):
        job = MixedJob(problems, models, loop=LOOP, concurrent=concurrent)
        for _ in range(2):
            got = job.run()
            torch.cuda.synchronize()
            assert all(torch.equal(a, b) for a, b in zip(got, s1)), concurrent
    # (e) family-aware shards (gnnmp.dist.shard_mixed: bucket by family, cut the family-major order at COST quantiles -- a
    # d = 64 kuka7 problem costs ~2.4 x a d = 32 one at equal edge count, so edge shares would be the wrong thing to balance):
    # every rank's shard gives the bytes the whole job gave, the shards partition the job, a rank holds at most two families
    # once there are as many ranks as families, and the predicted time of the slowest rank is within 10 % of the mean
    costs = problem_costs(problems, models, LOOP)
    for world in (2, 4, 8):
        seen = []
        loads = []
        for rank in range(world):
            idx = shard_mixed(problems, rank, world, models, LOOP)
            seen += idx
            loads.append(plan_times([idx], [p['env'] for p in problems], costs, family_time_curves(models))[0])
            if world >= 4:
                assert len({problems[i]['env'] for i in idx}) <= 2, (world, rank)
            if world == 4 or rank in (0, world - 1):                                    # scoring every shard of every split would only repeat (b)
                part = MixedJob([problems[i] for i in idx], models, loop=LOOP).run()
                torch.cuda.synchronize()
                assert all(torch.equal(a, s1[i]) for a, i in zip(part, idx)), (world, rank)
        assert sorted(seen) == list(range(len(problems)))
        assert max(loads) <= 1.10 * sum(loads) / world, (world, loads)
    # the contiguous edge-weighted split of the caller's order still works for single-family sets (tests/test_dist_gloo.py)
    lo, hi = shard_range(len(problems), 0, 2, [int(p['edge_index'].shape[1]) for p in problems])
    assert 0 < hi < len(problems)

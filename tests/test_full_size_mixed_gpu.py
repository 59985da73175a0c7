"""BASELINE configs[3] at FULL size on one GPU: the mixed environment set (maze / snake / ur5 / kuka, constructor sites
str2name.py:14,22,38,46 of the reference) -- 256 problems, 64 per family, 1000-node k1 = 8 RGGs, fp32 -- scored through
``gnnmp.dist.run_mixed`` (one batched forward per family, results in the caller's order).  Size-independent properties:
(a) two runs give identical bytes; (b) a graph scored inside the mixed job equals the same graph scored alone, bit for bit,
for every family; (c) one sampled graph per family agrees with the CPU oracle within the absolute fp32 bar of
tests/parity_bar.py; (d) the per-rank shard of the job (gnnmp.dist.shard_range weighted by edge counts) scores its problems
to the same bytes as the whole job does.  kuka7 (d = 64) runs the fp32 d = 64 kernels here -- a BASELINE config member."""
import pytest
import torch

from conftest import load_weights
import gnnmp
from gnnmp.dist import run_mixed, shard_range
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
    # (d) an edge-weighted 2-way split of the job: every rank's shard gives the bytes the whole job gave
    weights = [int(p['edge_index'].shape[1]) for p in problems]
    covered = 0
    for rank in range(2):
        lo, hi = shard_range(len(problems), rank, 2, weights)
        part = run_mixed(problems[lo:hi], models, loop=LOOP)
        assert all(torch.equal(a, b) for a, b in zip(part, s1[lo:hi]))
        covered += hi - lo
        share = sum(weights[lo:hi]) / sum(weights)
        assert 0.45 < share < 0.55, share
    assert covered == len(problems)

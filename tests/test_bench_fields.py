"""bench.py's bookkeeping that needs no GPU: the credited / executed split of `whole_forward` (no unlabelled fraction above 1), the
executed-FLOP entries of profiles/kernel_mfma.json being stamped with the sources they were measured on, and the workload list of
the `other_configs_gpu` leg being BASELINE configs[2] / [4] (str2name.py:46-64 shapes)."""
import json
import os

from conftest import REPO

import bench


def test_whole_forward_labels_credited_and_executed():
    wkey = 'maze2 N=1000 k1=8 graphs=256 fp32'
    blk = bench.whole_forward_block(wkey, flops_step=4.7e11, bytes_step=3.66e7, step_s=3.13e-3, is_bf16=False)
    # every fraction is either labelled credited, or executed, or is the HBM fraction of the algorithmic bytes
    fracs = {k: v for k, v in blk.items() if isinstance(v, float) and ('frac' in k)}
    assert set(fracs) <= {'credited_frac_fp32_peak', 'frac_executed', 'frac_hbm_peak'}
    assert 'not a utilisation' in blk['credited']
    ex = bench.executed_mfma(wkey)
    if ex is not None and not ex['stale']:
        assert 0.0 < blk['frac_executed'] < 1.0                      # counter-measured MFMA work can never exceed the peak
        assert blk['frac_executed'] <= blk['credited_frac_fp32_peak'] * 1.05 or blk['credited_frac_fp32_peak'] < blk['frac_executed']
    # a credited fraction MAY exceed 1 (fewer operations for the same result): it must still carry the label
    big = bench.whole_forward_block('no such workload', flops_step=1e12, bytes_step=1.0, step_s=1e-3, is_bf16=False)
    assert big['credited_frac_fp32_peak'] > 1.0 and big['executed_TFLOPs'] is None and big['frac_executed'] is None


def test_kernel_mfma_entries_are_stamped_and_consistent():
    path = os.path.join(REPO, 'profiles', 'kernel_mfma.json')
    ents = json.load(open(path))
    keys = {e['workload'] for e in ents}
    assert {'maze2 N=1000 k1=8 graphs=256 fp32', 'kuka7 N=2000 k1=10 graphs=64 bf16', 'kuka14 N=5000 k1=16 graphs=32 bf16'} <= keys
    for e in ents:
        assert len(e['kernel_source_sha256']) == 64 and e['steps_in_run'] > 0
        total = sum(k['mfma_flops_per_launch'] * k['launches_per_step'] for k in e['kernels'])
        assert abs(total - e['executed_mfma_flops_per_step']) <= 1e-6 * total
        # the message-passing kernel runs loop = 5 times per step, everything else once
        for k in e['kernels']:
            assert k['launches_per_step'] == (5.0 if 'mp_fused' in k['kernel'] else 1.0), k['kernel']


def test_other_configs_are_the_baseline_bf16_shapes():
    base = json.load(open(os.path.join(REPO, 'BASELINE.json')))['configs']
    assert '2000-node k=10' in base[2] and 'bf16' in base[2] and '5000-node k=16' in base[4] and 'bf16' in base[4]
    got = {(env, n, k, dt) for _, env, n, k, _, dt in bench.OTHER_CONFIGS}
    assert got == {('kuka7', 2000, 10, 'bf16'), ('kuka14', 5000, 16, 'bf16')}

"""Pin the CPU oracle (oracle/ref_cpu.py) to the golden vectors recorded from the unmodified
reference (tools/gen_golden.py): every shipped explorer and smoother checkpoint, fp32 and
fp64, plus intermediate activations.  CPU only."""
import os

import numpy as np
import pytest
import torch

from conftest import env_of, golden_files, load_weights
from oracle import ref_cpu
import gnnmp  # noqa: F401
from gnnmp.synth import ENVS


def _load(path):
    with np.load(path) as f:
        return {k: f[k] for k in f.files}


# max|oracle (default form) - golden fp32| measured per fixture in the authoring container (tools: see DESIGN.md section 2)
_PLAIN_GAP = {
    'explorer_kuka13_N64_k4_L5.npz': 5.13e-06, 'explorer_kuka14_N200_k8_L5.npz': 4.05e-06, 'explorer_kuka14_N64_k4_L5.npz': 4.29e-06,
    'explorer_kuka7_N200_k6_L5.npz': 2.86e-06, 'explorer_kuka7_N64_k4_L2_noobs.npz': 0.0, 'explorer_kuka7_N64_k4_L5.npz': 2.86e-06,
    'explorer_maze2_N1000_k8_L5.npz': 1.14e-05, 'explorer_maze2_N200_k6_L5.npz': 2.77e-05, 'explorer_maze2_N64_k4_L1.npz': 7.63e-06,
    'explorer_maze2_N64_k4_L3.npz': 7.63e-06, 'explorer_maze2_N64_k4_L5.npz': 7.63e-06, 'explorer_maze2_N64_k4_L5_noobs.npz': 0.0,
    'explorer_maze3_N64_k4_L5.npz': 5.72e-06, 'explorer_snake7_N64_k4_L5.npz': 5.72e-06, 'explorer_ur5_N64_k4_L5.npz': 5.72e-06,
}


@pytest.mark.parametrize('path', golden_files('explorer_'), ids=os.path.basename)
def test_explorer_oracle_matches_reference(path):
    r = _load(path)
    env = env_of(path)
    w = load_weights(ENVS[env]['ckpt'])
    args = dict(v=torch.from_numpy(r['v']), goal=torch.from_numpy(r['goal']),
                obstacles=torch.from_numpy(r['obstacles']),
                edge_index=torch.from_numpy(r['edge_index']), loop=int(r['loop']),
                use_obstacles=bool(r['use_obstacles']))
    taps = {}
    s = ref_cpu.explorer_forward(w, taps=taps, **args)
    ref32 = torch.from_numpy(r['scores_fp32'])
    # the oracle's default (contracted value-mix) form rounds differently from the reference's materialising form: held to the
    # distance MEASURED per fixture (round 4, identical at 1 and 8 torch threads) + 25 % + 1e-6 for BLAS blocking on other hosts
    # -- never looser than the bar the GPU is held to against the same golden (tests/parity_bar.py)
    gap = _PLAIN_GAP[os.path.basename(path)]
    assert (s - ref32).abs().max().item() <= 1.25 * gap + 1e-6, ((s - ref32).abs().max().item(), gap)
    # fp64 run of the oracle reproduces the fp64 run of the reference to fp64 roundoff
    w64 = {k: v.double() for k, v in w.items()}
    a64 = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in args.items()}
    s64 = ref_cpu.explorer_forward(w64, **a64)
    assert torch.allclose(s64, torch.from_numpy(r['scores_fp64']), rtol=1e-10, atol=1e-10)
    # the materialising attention form (the reference's literal one) agrees too
    # the materialising attention form (the reference's literal one, model.py:178-179) is the PIN: it reproduces the reference's
    # fp32 run bit for bit at the thread count the goldens were recorded with (8) and to <= 3.8e-6 at one thread (BLAS blocking);
    # bar = that worst case + 25 %
    sm = ref_cpu.explorer_forward(w, materialize=True, **args)
    assert (sm - ref32).abs().max().item() <= 4.8e-6, (sm - ref32).abs().max().item()
    if 'tap_node_code' in r:
        for key, mine in (('tap_node_code', taps['node_code']), ('tap_edge_code', taps['edge_code']),
                          ('tap_node_free_code', taps['node_free_code']),
                          ('tap_edge_free_code', taps['edge_free_code']),
                          ('tap_h', torch.stack(taps['h'])), ('tap_decode', taps['decode'])):
            assert torch.allclose(mine, torch.from_numpy(r[key]), rtol=1e-4, atol=2e-5), key


def test_explorer_dense_orientation():
    """P[target, source] (model.py:148-149), zero elsewhere."""
    r = _load(golden_files('explorer_maze2_N64_k4_L5.npz')[0])
    w = load_weights('weights_maze')
    ei = torch.from_numpy(r['edge_index'])
    kw = dict(v=torch.from_numpy(r['v']), goal=torch.from_numpy(r['goal']),
              obstacles=torch.from_numpy(r['obstacles']), edge_index=ei, loop=5)
    P = ref_cpu.explorer_forward(w, dense=True, **kw)
    s = ref_cpu.explorer_forward(w, **kw)
    assert torch.equal(P[ei[1], ei[0]], s)
    assert int((P != 0).sum()) <= ei.shape[1]


def test_explorer_loop_zero_raises():
    r = _load(golden_files('explorer_maze2_N64_k4_L5.npz')[0])
    w = load_weights('weights_maze')
    with pytest.raises(ValueError):
        ref_cpu.explorer_forward(w, torch.from_numpy(r['v']), torch.from_numpy(r['goal']),
                                 torch.from_numpy(r['obstacles']), torch.from_numpy(r['edge_index']), 0)


@pytest.mark.parametrize('path', golden_files('smoother_'), ids=os.path.basename)
def test_smoother_oracle_matches_reference(path):
    r = _load(path)
    name = os.path.basename(path).split('_P')[0].replace('smoother_', '')
    w = load_weights(name)
    kw = dict(path=torch.from_numpy(r['path']), free=torch.from_numpy(r['free']),
              collided=torch.from_numpy(r['collided']), edge_index=torch.from_numpy(r['edge_index']),
              loop=int(r['loop']), scale=float(r['scale']))
    keep = kw['path'].clone()
    if 'out_fp32_knn64' in r:
        # the float32-kNN fixture (tools/gen_golden.py smoother_knn32_case): the oracle with input-dtype distances
        # reproduces the run recorded under the float32 stand-in, the float64-distance oracle the other run, and the
        # two differ by far more than the parity bar -- i.e. the fixture does discriminate the kNN dtype
        out32 = ref_cpu.smoother_forward(w, knn_input_dtype=True, **kw)
        out64k = ref_cpu.smoother_forward(w, **kw)
        assert torch.allclose(out32, torch.from_numpy(r['out_fp32']), rtol=1e-5, atol=1e-5)
        assert torch.allclose(out64k, torch.from_numpy(r['out_fp32_knn64']), rtol=1e-5, atol=1e-5)
        assert (out32 - out64k).abs().max() > 1e-4
        return
    out = ref_cpu.smoother_forward(w, **kw)
    assert torch.equal(keep, kw['path'])            # caller's path untouched
    assert torch.allclose(out, torch.from_numpy(r['out_fp32']), rtol=1e-5, atol=1e-5)
    w64 = {k: v.double() for k, v in w.items()}
    kw64 = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in kw.items()}
    out64 = ref_cpu.smoother_forward(w64, **kw64)
    assert torch.allclose(out64, torch.from_numpy(r['out_fp64']), rtol=1e-10, atol=1e-10)


def test_primitives_edge_cases():
    # empty target rows aggregate to 0 (torch_scatter semantics), not -inf
    msg = torch.tensor([[-3.0, -1.0], [-2.0, -5.0]])
    out = ref_cpu.scatter_rows(msg, torch.tensor([2, 2]), 4, 'max')
    assert torch.equal(out, torch.tensor([[0., 0.], [0., 0.], [-2., -1.], [0., 0.]]))
    out = ref_cpu.scatter_rows(msg, torch.tensor([0, 0]), 2, 'add')
    assert torch.equal(out, torch.tensor([[-5., -6.], [0., 0.]]))
    # coalesce sorts by (row0,row1) and removes duplicates
    e = torch.tensor([[2, 0, 2, 1], [1, 3, 1, 0]])
    assert torch.equal(ref_cpu.coalesce(e, 4), torch.tensor([[0, 1, 2], [3, 0, 1]]))
    # knn_graph(loop=True) includes the point itself and orients (neighbour -> centre)
    x = torch.tensor([[0.0], [1.0], [3.0]])
    g = ref_cpu.knn_graph(x, 2, loop=True)
    pairs = set(map(tuple, g.T.tolist()))
    assert pairs == {(0, 0), (1, 0), (1, 1), (0, 1), (2, 2), (1, 2)}
    # fewer points than k
    assert ref_cpu.knn(x, x[:1], 10).shape == (2, 3)


def test_graph_build_matches_oracle():
    """The product's host graph builder (graph_build.py, row H1) == the oracle's."""
    from gnnmp import graph_build
    gen = torch.Generator().manual_seed(7)
    v = torch.rand(90, 3, generator=gen)
    a = graph_build.build_edges(v, 40, 5)
    b = ref_cpu.build_edges(v, 40, 5)
    assert torch.equal(a, b)
    assert graph_build.k1_of(30, 502) == ref_cpu.k1_of(30, 502) == 41

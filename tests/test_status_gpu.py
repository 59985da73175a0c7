"""Device-side status of a forward (include/gnnmp.h: gnnmp_*_status): what used to be silent wrong answers is an error.

The reference attends over ALL obstacles it is given (model.py:125-130) and its tensor indexing raises on a node id outside
the graph; the library sizes its K/V slabs / LDS carve-ups from caller promises (max_obstacles, max_path, max_samples,
max_edges) that live next to DEVICE-resident prefix arrays it never reads back.  A forward that finds a promise broken, or a
node id out of range, writes that into a status region of the workspace; the C ABI reads it with gnnmp_*_status (blocking),
the Python wrappers copy it to the host behind the forward and raise on a later call or in check_status()."""
import ctypes

import pytest
import torch

import gnnmp
from gnnmp import _lib
from gnnmp.smoother import SmoothBatch
from gnnmp.planner import chain_edge_index
from gnnmp.synth import ENVS, synth_graph
from conftest import load_weights

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _model(env='maze2'):
    e = ENVS[env]
    m = gnnmp.EncoderProcessDecoder(e['workspace'], e['C'], e['d'], e['S']).eval()
    m.load_state_dict(load_weights(e['ckpt']), strict=True)
    return m


def test_more_obstacles_than_promised_is_an_error():
    m = _model()
    graphs = [synth_graph('maze2', 64, 4, seed=1), synth_graph('maze2', 40, 3, seed=2)]
    good = gnnmp.GraphBatch.from_graphs(graphs, 2, DEV)
    n_obs = [int(g['obstacles'].reshape(-1, 2).shape[0]) for g in graphs]
    assert max(n_obs) > 32
    s_good = m.forward_batch(good, 2)
    m.check_status()                                              # nothing to report
    # the same batch with a promise of 32 obstacles per graph: the slabs hold one 32-obstacle tile, the attention sees the first
    # 32 only -- different scores, and the status says so
    short = gnnmp.GraphBatch(good.v, good.goal, good.obstacles, good.edge_index, good.node_ptr, good.edge_ptr, good.obs_ptr, 32,
                             dense_floats=good.dense_floats)
    s_short = m.forward_batch(short, 2)
    assert not torch.equal(s_good, s_short)
    with pytest.raises(RuntimeError, match='max_obstacles'):
        m.check_status()
    # raw C ABI, blocking (a forward WITHOUT a status slot leaves the words in its workspace): GNNMP_ERR_CAPS (-7) and the first
    # offending graph
    m.status_checks = False
    assert torch.equal(m.forward_batch(short, 2), s_short)
    cb = m._cbatch(short)
    first = ctypes.c_int32(-5)
    rc = _lib.lib().gnnmp_explorer_status(m._native(torch.device(DEV)), ctypes.byref(cb), m._ws.data_ptr(), m._ws.numel(), None,
                                          ctypes.byref(first))
    assert rc == -7 and first.value == min(i for i, n in enumerate(n_obs) if n > 32)
    with pytest.raises(RuntimeError, match='max_obstacles'):
        m.check_status(short)
    m.forward_batch(good, 2)
    m.check_status(good)                                          # the blocking form on a clean forward
    m.status_checks = True
    # non-blocking path: the NEXT forward on the module raises once the earlier forward's status copy has arrived
    m.forward_batch(short, 2)
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match='max_obstacles'):
        m.forward_batch(good, 2)
    # and a clean forward afterwards is clean (every slot is rewritten by every forward: nothing sticks)
    assert torch.equal(m.forward_batch(good, 2), s_good)
    m.check_status()
    # use_obstacles = False ignores the obstacles altogether (model.py:125): no promise to break
    m.use_obstacles = False
    m.forward_batch(short, 2)
    m.check_status()


@pytest.mark.parametrize('nodes,k', [(64, 4), (9000, 8)], ids=['one_launch_prep', 'split_prep'])
def test_node_id_outside_its_graph_is_an_error(nodes, k):
    """Both prep forms (one launch; histogram + scatter for graphs of > ~12 k edges): an id outside [0, N_g) is replaced by node 0
    (nothing reads or writes outside the graph's rows) and reported as GNNMP_ERR_INDEX."""
    m = _model()
    graphs = [synth_graph('maze2', nodes, k, seed=3), synth_graph('maze2', 40, 3, seed=4)]
    b = gnnmp.GraphBatch.from_graphs(graphs, 2, DEV)
    clean = m.forward_batch(b, 2)
    m.check_status()
    for row, col, val in ((0, -1, 40), (1, 0, -1), (1, 5, nodes)):            # source of the last graph, targets of the first
        bad = gnnmp.GraphBatch.from_graphs(graphs, 2, DEV)
        bad.edge_index[row, col] = val
        s = m.forward_batch(bad, 2)
        assert bool(torch.isfinite(s).all())
        with pytest.raises(RuntimeError, match='node ids'):
            m.check_status()
        m.status_checks = False                                   # raw blocking ABI: words in the workspace
        assert torch.equal(m.forward_batch(bad, 2), s)
        m.status_checks = True
        first = ctypes.c_int32(-5)
        cb = m._cbatch(bad)
        rc = _lib.lib().gnnmp_explorer_status(m._native(torch.device(DEV)), ctypes.byref(cb), m._ws.data_ptr(), m._ws.numel(), None,
                                              ctypes.byref(first))
        assert rc == -8 and first.value == (1 if col == -1 else 0)
    assert torch.equal(m.forward_batch(b, 2), clean)
    m.check_status()


def test_status_checks_can_be_switched_off():
    m = _model()
    m.status_checks = False
    g = synth_graph('maze2', 64, 4, seed=1)
    b = gnnmp.GraphBatch.from_graphs([g], 2, DEV)
    short = gnnmp.GraphBatch(b.v, b.goal, b.obstacles, b.edge_index, b.node_ptr, b.edge_ptr, b.obs_ptr, 32, dense_floats=b.dense_floats)
    m.forward_batch(short, 2)
    m.check_status()                                              # no copies were made: nothing pending
    with pytest.raises(RuntimeError):                             # the blocking form still sees the device-side words
        m.check_status(short)


def _smoother():
    ms = gnnmp.ModelSmoother(workspace_size=2, config_size=2, embed_size=128, obs_size=6).eval()
    ms.load_state_dict(load_weights('smooth_2d_attv3'))
    return ms


def test_smoothing_problem_beyond_its_caps_is_an_error():
    ms = _smoother()
    gen = torch.Generator().manual_seed(5)
    paths = [torch.rand(n, 2, generator=gen) * 2 - 1 for n in (12, 20)]
    frees = [torch.rand(n, 2, generator=gen) * 2 - 1 for n in (60, 70)]
    colls = [torch.rand(n, 2, generator=gen) * 2 - 1 for n in (40, 45)]
    eis = [chain_edge_index(12), chain_edge_index(20)]
    sb = SmoothBatch(paths, frees, colls, eis, DEV)
    good = ms.forward_batch(sb, 1)
    ms.check_status()
    ms.check_status(sb)                                           # caps from host-side counts: no slot was used, the blocking form reads the workspace
    for field, value in (('max_path', 12), ('max_samples', 100), ('max_edges', int(eis[0].shape[1]))):       # each fits problem 0 only
        lie = SmoothBatch(paths, frees, colls, eis, DEV)
        setattr(lie, field, value)
        lie.caps_from_host = False                                # a hand-set cap: the wrapper copies the status word back
        out = ms.forward_batch(lie, 1)
        assert torch.equal(out[:12], good[:12])                   # the problem inside the caps is untouched
        assert not torch.equal(out[12:], good[12:])               # the one beyond them lost its edges
        with pytest.raises(RuntimeError, match='max_path / max_samples / max_edges'):
            ms.check_status()
        ms.status_checks = 'never'                                # raw blocking ABI: words in the workspace
        assert torch.equal(ms.forward_batch(lie, 1), out)
        ms.status_checks = 'auto'
        first = ctypes.c_int32(-5)
        from gnnmp.smoother import _cbatch
        cb = _cbatch(lie)
        rc = _lib.lib().gnnmp_smoother_status(ms._native(torch.device(DEV)), ctypes.byref(cb), ms._ws.data_ptr(), ms._ws.numel(), None,
                                              ctypes.byref(first))
        assert rc == -7 and first.value == 1
    assert torch.equal(ms.forward_batch(sb, 1), good)
    ms.check_status(sb)


def test_single_graph_call_reports_bad_ids():
    """The reference's own call shape (ONE graph, eval_gnn.py:194): an out-of-range edge_index id is clamped by the kernels (finite
    scores) and must still be reported -- the reference's indexing raises there."""
    m = _model()
    g = synth_graph('maze2', 64, 4, seed=1)
    kw = dict(goal=g['goal'].to(DEV), v=g['v'].to(DEV), obstacles=g['obstacles'].to(DEV), loop=2)
    ei = g['edge_index'].to(DEV)
    P = m(edge_index=ei, **kw)
    m.check_status()
    bad = ei.clone()
    bad[0, 3] = 64
    Pb = m(edge_index=bad, **kw)
    assert bool(torch.isfinite(Pb).all())
    with pytest.raises(RuntimeError, match='node ids'):
        m.check_status()
    sb = m.edge_scores(edge_index=bad, **kw)
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match='node ids'):
        m(edge_index=ei, **kw)                                    # the next forward sees the arrived slot
    assert torch.equal(m(edge_index=ei, **kw), P)
    m.check_status()


def test_status_ring_survives_many_forwards_and_keeps_unfinished_slots():
    """More forwards than ring slots without ever asking: the oldest slot is waited for and recycled; an error among them
    surfaces, later clean forwards stay clean."""
    m = _model()
    graphs = [synth_graph('maze2', 64, 4, seed=1), synth_graph('maze2', 40, 3, seed=2)]
    good = gnnmp.GraphBatch.from_graphs(graphs, 2, DEV)
    short = gnnmp.GraphBatch(good.v, good.goal, good.obstacles, good.edge_index, good.node_ptr, good.edge_ptr, good.obs_ptr, 32,
                             dense_floats=good.dense_floats)
    ref = m.forward_batch(good, 2)
    m.check_status()
    for _ in range(100):
        assert torch.equal(m.forward_batch(good, 2), ref)
    m.check_status()
    watch = m._status_watch()
    assert len(watch.free) == watch.depth and not watch.order
    m.forward_batch(short, 2)
    with pytest.raises(RuntimeError, match='max_obstacles'):
        for _ in range(100):
            m.forward_batch(good, 2)
    m.check_status()                                              # the clean forwards before the raise left nothing behind
    assert len(watch.free) == watch.depth

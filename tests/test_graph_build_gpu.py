"""Device-side graph construction (csrc/graph_kernels.hip) against the host
builder / oracle: the coalesced edge_index must be IDENTICAL (integer work: bit-exact bar)."""
import numpy as np
import pytest
import torch

from conftest import golden_files
import gnnmp
from gnnmp import graph_build
from oracle import ref_cpu

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _batch(specs, seed):
    gen = torch.Generator().manual_seed(seed)
    vs, nf, k1 = [], [], []
    for n, f, k, c in specs:
        vs.append(torch.rand(n, c, generator=gen) * 2 - 1)
        nf.append(f)
        k1.append(k)
    return vs, nf, k1


def _check(vs, nf, k1):
    C = vs[0].shape[1]
    ptr = torch.zeros(len(vs) + 1, dtype=torch.int32)
    ptr[1:] = torch.tensor([v.shape[0] for v in vs]).cumsum(0).to(torch.int32)
    ei, eptr = graph_build.build_edges_gpu(torch.cat(vs).to(DEV), ptr.to(DEV), nf, k1)
    eptr = eptr.cpu().tolist()
    for g, v in enumerate(vs):
        ref = ref_cpu.build_edges(v, nf[g], k1[g])
        mine = ei[:, eptr[g]:eptr[g + 1]].cpu()
        assert mine.shape == ref.shape, (g, mine.shape, ref.shape)
        assert torch.equal(mine, ref), g
        assert torch.equal(ref, graph_build.build_edges(v, nf[g], k1[g]))
    assert C == vs[0].shape[1]


def test_matches_host_builder_ragged_batch():
    vs, nf, k1 = _batch([(200, 100, 6, 2), (64, 32, 4, 2), (333, 150, 9, 2), (1000, 500, 8, 2), (5, 2, 3, 2)], 1)
    _check(vs, nf, k1)


@pytest.mark.parametrize('C', [3, 7, 14])
def test_higher_dimensional_configs(C):
    vs, nf, k1 = _batch([(150, 70, 7, C), (90, 45, 16, C)], C)
    _check(vs, nf, k1)


def test_degenerate_k():
    # k larger than the free set and larger than the whole graph (knn clamps to the set size)
    gen = torch.Generator().manual_seed(9)
    _check([torch.rand(12, 2, generator=gen), torch.rand(3, 2, generator=gen)], [4, 3], [6, 8])


def test_exact_ties_pick_lower_index():
    """Exact distance ties at the k-th neighbour are unspecified in the reference's kNN;
    the device builder resolves them by the lower index."""
    ptr = torch.tensor([0, 3], dtype=torch.int32, device=DEV)
    tie = torch.tensor([[0.0, 0.0], [1.0, 0.0], [-1.0, 0.0]])
    ei, _ = graph_build.build_edges_gpu(tie.to(DEV), ptr, [3], [2])
    pairs = set(map(tuple, ei.cpu().T.tolist()))
    assert (1, 0) in pairs and (0, 1) in pairs          # node 0 picked index 1 (lower) over index 2 ...
    assert (2, 0) in pairs and (0, 2) in pairs          # ... but node 2's own list brings (0 -> 2) and its reverse


def test_reference_trace_graphs():
    """The graphs the reference planner built on real maze problems (recorded create_data outputs)."""
    for path in golden_files('planner_'):
        with np.load(path) as f:
            v = torch.from_numpy(f['e0_v'])
            ref = torch.from_numpy(f['e0_edge_index'])
            n_free = int(f['e0_free'].shape[0])
            k1 = graph_build.k1_of(int(f['k']), n_free)
        ptr = torch.tensor([0, v.shape[0]], dtype=torch.int32, device=DEV)
        ei, _ = graph_build.build_edges_gpu(v.to(DEV), ptr, [n_free], [k1])
        assert torch.equal(ei.cpu(), ref)


def test_feeds_the_explorer():
    """edge_index built on the GPU goes straight into the explorer (no host round trip of the graph)."""
    from conftest import load_weights
    from gnnmp.synth import synth_graph
    g = synth_graph('maze2', 300, 6, seed=3)
    ptr = torch.tensor([0, 300], dtype=torch.int32, device=DEV)
    ei, _ = graph_build.build_edges_gpu(g['v'].to(DEV), ptr, [150], [6])
    assert torch.equal(ei.cpu(), g['edge_index'])
    m = gnnmp.EncoderProcessDecoder(2, 2, 32, 2).eval()
    m.load_state_dict(load_weights('weights_maze'))
    s = m.edge_scores(g['goal'].to(DEV), 5, g['v'].to(DEV), g['obstacles'].to(DEV), ei).cpu()
    from parity_bar import assert_fp32_parity, explorer_oracle_pair
    ref32, ref64 = explorer_oracle_pair(load_weights('weights_maze'), g, 5)
    assert_fp32_parity(s, ref32, ref64, 'device-built graph')


def test_hub_bucket_beyond_lds_share_and_large_graphs():
    """(a) a hub that is the nearest neighbour of every other node: its per-source bucket (N entries) exceeds the
    wave's LDS share in gb_unique (cap = 256 at k1 = 2) and takes the in-place fallback; (b) graphs of 1500 and
    2500 nodes: the 32-per-lane register cache and the general recomputing path of gb_knn."""
    gen = torch.Generator().manual_seed(21)
    x = torch.randn(600, 7, generator=gen)
    x = x / x.norm(dim=1, keepdim=True)            # points on the unit sphere: pairwise ~ sqrt(2), hub at distance 1
    x[5] = 0.0
    _check([x, torch.rand(50, 7, generator=gen)], [400, 20], [2, 3])
    _check([torch.rand(1500, 2, generator=gen) * 2 - 1], [700], [5])
    _check([torch.rand(2500, 3, generator=gen) * 2 - 1], [1200], [4])          # beyond 2048 nodes, k <= 16: streamed per-lane top-16
    _check([torch.rand(2300, 2, generator=gen) * 2 - 1], [1100], [19])         # ... k > 16: the form that recomputes the distances per round
    # small and large graphs in one batch: the first kNN launch handles the small ones and lists the others' nodes for the second
    _check([torch.rand(300, 2, generator=gen), torch.rand(1500, 2, generator=gen), torch.rand(40, 2, generator=gen),
            torch.rand(2100, 2, generator=gen), torch.rand(1024, 2, generator=gen)], [100, 800, 40, 1000, 500], [4, 5, 3, 4, 6])


def test_small_capacity_is_retried(monkeypatch):
    """Far fewer columns than needed on the first try (duplicate-free worst case: k1 = 1 gives only self loops, so make
    the estimate too small by shrinking it): the builder repeats with the exact size and returns the same edges."""
    import builtins
    gen = torch.Generator().manual_seed(4)
    vs = [torch.rand(1500, 2, generator=gen), torch.rand(400, 2, generator=gen)]       # ~20 k columns, first try: 4112
    real_int = builtins.int
    # force the first estimate down to a handful of columns
    monkeypatch.setattr(graph_build, 'int', lambda x: 16 if isinstance(x, float) else real_int(x), raising=False)
    _check(vs, [700, 200], [8, 6])

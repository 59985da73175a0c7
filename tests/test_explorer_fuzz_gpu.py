"""Randomised shapes against the CPU oracle: arbitrary edge lists (not kNN: duplicates, missing self loops,
isolated nodes, hubs whose incoming edges span several 32-edge tiles), ragged batches, random obstacle counts
and loop counts.  Exercises the CSR build, the segmented max across tile boundaries (part_first / part_last)
and the padded index spaces far away from the benchmark's regular shapes."""
import pytest
import torch

from conftest import load_weights
import gnnmp
from oracle import ref_cpu
from parity_bar import assert_fp32_parity

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def random_graph(gen, n, e, n_obs, hub=None):
    v = torch.rand(n, 2, generator=gen) * 2 - 1
    src = torch.randint(0, n, (e,), generator=gen)
    dst = torch.randint(0, n, (e,), generator=gen)
    if hub is not None:                                  # a hub collects `hub` incoming edges (several tiles long)
        k = min(hub, e)
        dst[:k] = int(torch.randint(0, n, (1,), generator=gen))
    ei = torch.stack((src, dst))
    obstacles = torch.rand(n_obs, 2, generator=gen) - 0.5
    return {'v': v, 'goal': v[int(torch.randint(0, n, (1,), generator=gen))].clone(), 'obstacles': obstacles, 'edge_index': ei}


def oracle(w, g, loop, dtype=torch.float32):
    wd = {k: (t.to(dtype) if t.is_floating_point() else t) for k, t in w.items()}
    return ref_cpu.explorer_forward(wd, g['v'].to(dtype), g['goal'].to(dtype), g['obstacles'].to(dtype), g['edge_index'], loop)


def check(part, w, g, loop):
    """Structure fuzz: the same per-input fp32 bar as the goldens (tests/parity_bar.py)."""
    ref32, ref64 = oracle(w, g, loop), oracle(w, g, loop, torch.float64)
    r = assert_fp32_parity(part, ref32, ref64, 'fuzz')
    return r['err32'], r['err64'], r['own']


@pytest.mark.parametrize('seed', range(12))
def test_random_graphs(seed):
    gen = torch.Generator().manual_seed(1000 + seed)
    w = load_weights('weights_maze')
    m = gnnmp.EncoderProcessDecoder(2, 2, 32, 2).eval()
    m.load_state_dict(w)
    graphs = []
    for i in range(int(torch.randint(1, 6, (1,), generator=gen))):
        n = int(torch.randint(1, 300, (1,), generator=gen))
        e = int(torch.randint(0, 6 * n + 2, (1,), generator=gen))
        n_obs = int(torch.randint(0, 140, (1,), generator=gen))
        hub = int(torch.randint(33, 200, (1,), generator=gen)) if (seed + i) % 2 == 0 and e > 40 else None
        graphs.append(random_graph(gen, n, e, n_obs, hub))
    loop = int(torch.randint(1, 7, (1,), generator=gen))
    b = gnnmp.GraphBatch.from_graphs(graphs, 2, DEV)
    out = m.forward_batch(b, loop)
    for g, part in zip(graphs, b.split_edges(out)):
        if g['edge_index'].shape[1] == 0:
            assert part.numel() == 0
            continue
        print('N=%d E=%d: err32 %.2e err64 %.2e own %.2e' % ((g['v'].shape[0], g['edge_index'].shape[1]) + check(part.cpu(), w, g, loop)))


def test_star_graph_hub_spans_many_tiles():
    gen = torch.Generator().manual_seed(5)
    n = 260
    v = torch.rand(n, 2, generator=gen) * 2 - 1
    src = torch.arange(1, n)
    ei = torch.cat((torch.stack((src, torch.zeros(n - 1, dtype=torch.long))),          # everyone -> node 0 (259 in-edges)
                    torch.stack((torch.zeros(n - 1, dtype=torch.long), src))), dim=1)  # node 0 -> everyone
    g = {'v': v, 'goal': v[3].clone(), 'obstacles': torch.rand(20, 2, generator=gen) - 0.5, 'edge_index': ei}
    w = load_weights('weights_maze')
    m = gnnmp.EncoderProcessDecoder(2, 2, 32, 2).eval()
    m.load_state_dict(w)
    s = m.edge_scores(g['goal'].to(DEV), 4, g['v'].to(DEV), g['obstacles'].to(DEV), ei.to(DEV)).cpu()
    print(check(s, w, g, 4))
    m2 = gnnmp.EncoderProcessDecoder(3, 7, 64, 6).eval()                                         # d = 64 fallback kernels too
    w2 = load_weights('weights_kuka')
    m2.load_state_dict(w2)
    g2 = dict(g, v=torch.rand(n, 7, generator=gen) * 2 - 1, obstacles=torch.rand(4, 6, generator=gen))
    g2['goal'] = g2['v'][9].clone()
    s2 = m2.edge_scores(g2['goal'].to(DEV), 3, g2['v'].to(DEV), g2['obstacles'].to(DEV), ei.to(DEV)).cpu()
    print(check(s2, w2, g2, 3))


def test_per_graph_csr_build_ragged_batch():
    """80 random graphs incl. empty edge lists, single nodes and hubs in one batch (one workgroup per graph in the CSR
    build, tile-per-wave message kernel) and bit-equality with the same graphs scored one by one (tile-per-workgroup
    message kernel, both pre stages in one launch)."""
    gen = torch.Generator().manual_seed(31)
    w = load_weights('weights_maze')
    m = gnnmp.EncoderProcessDecoder(2, 2, 32, 2).eval()
    m.load_state_dict(w)
    graphs = []
    for i in range(80):
        n = int(torch.randint(1, 120, (1,), generator=gen))
        e = 0 if i % 13 == 0 else int(torch.randint(1, 5 * n + 2, (1,), generator=gen))
        graphs.append(random_graph(gen, n, e, int(torch.randint(0, 40, (1,), generator=gen)), hub=40 if i % 7 == 3 and e > 50 else None))
    b = gnnmp.GraphBatch.from_graphs(graphs, 2, DEV)
    parts = b.split_edges(m.forward_batch(b, 3))
    for i, (g, part) in enumerate(zip(graphs, parts)):
        if g['edge_index'].shape[1] == 0:
            assert part.numel() == 0
            continue
        one = m.edge_scores(g['goal'].to(DEV), 3, g['v'].to(DEV), g['obstacles'].to(DEV), g['edge_index'].to(DEV))
        assert torch.equal(part, one), i
        if i % 8 == 0:
            check(part.cpu(), w, g, 3)


def test_graph_beyond_the_lds_share_of_the_csr_build():
    """9 000 nodes inside a 66-graph batch: the CSR build keeps this graph's counters in global memory instead of LDS
    (kPrepCap = 8192), the 65 small graphs next to it take the LDS path."""
    gen = torch.Generator().manual_seed(77)
    w = load_weights('weights_maze')
    m = gnnmp.EncoderProcessDecoder(2, 2, 32, 2).eval()
    m.load_state_dict(w)
    graphs = [random_graph(gen, 30, 100, 7) for _ in range(30)] + [random_graph(gen, 9000, 30000, 20, hub=150)] + \
        [random_graph(gen, 50, 200, 7) for _ in range(35)]
    b = gnnmp.GraphBatch.from_graphs(graphs, 2, DEV)
    parts = b.split_edges(m.forward_batch(b, 2))
    for i in (0, 30, 31, 65):
        print(check(parts[i].cpu(), w, graphs[i], 2))


def test_sliced_csr_build_small_batches():
    """Few small graphs: the CSR build cuts a graph's target nodes into slices, one workgroup each (one to eight by batch size).
    (a) batches of 1, 3, 40 and 300 graphs, bit-equal to the same graphs scored one by one; (b) three graphs of which one has
    more than 16 k edges: its slices take the form that keeps the ranks in the workspace instead of registers."""
    w = load_weights('weights_maze')
    m = gnnmp.EncoderProcessDecoder(2, 2, 32, 2).eval()
    m.load_state_dict(w)
    gen = torch.Generator().manual_seed(5)
    for count in (1, 3, 40, 300):
        graphs = [random_graph(gen, int(torch.randint(130, 700, (1,), generator=gen)), int(torch.randint(200, 3000, (1,), generator=gen)),
                               int(torch.randint(0, 30, (1,), generator=gen))) for _ in range(count)]
        b = gnnmp.GraphBatch.from_graphs(graphs, 2, DEV)
        parts = b.split_edges(m.forward_batch(b, 2))
        for i in range(0, count, max(1, count // 6)):
            g = graphs[i]
            one = m.edge_scores(g['goal'].to(DEV), 2, g['v'].to(DEV), g['obstacles'].to(DEV), g['edge_index'].to(DEV))
            assert torch.equal(parts[i], one), (count, i)
        print(check(parts[0].cpu(), w, graphs[0], 2))
    graphs = [random_graph(gen, 1000, 17500, 12), random_graph(gen, 150, 400, 5), random_graph(gen, 300, 900, 9)]
    b = gnnmp.GraphBatch.from_graphs(graphs, 2, DEV)
    parts = b.split_edges(m.forward_batch(b, 2))
    for i, g in enumerate(graphs):
        one = m.edge_scores(g['goal'].to(DEV), 2, g['v'].to(DEV), g['obstacles'].to(DEV), g['edge_index'].to(DEV))
        assert torch.equal(parts[i], one), i
    print(check(parts[1].cpu(), w, graphs[1], 2))


@pytest.mark.parametrize('seed', range(4))
@pytest.mark.parametrize('mode', ['kuka7_fp32', 'maze_bf16x3', 'kuka7_bf16'])
def test_random_graphs_other_kernels(mode, seed):
    """The same structure fuzz through the d = 64 kernels (kuka7 checkpoint: general pre_kernel, two feature tiles),
    the bf16x3 kernels (held to the fp32 bar) and the bf16 kernels (against the exact-fp32 GPU result, bf16 bar)."""
    gen = torch.Generator().manual_seed(4000 + seed)
    ck, C, d, S, ws = ('weights_maze', 2, 32, 2, 2) if mode.startswith('maze') else ('weights_kuka', 7, 64, 6, 3)
    w = load_weights(ck)
    m = gnnmp.EncoderProcessDecoder(ws, C, d, S).eval()
    m.load_state_dict(w)
    graphs = []
    for i in range(int(torch.randint(2, 5, (1,), generator=gen))):
        n = int(torch.randint(2, 220, (1,), generator=gen))
        e = int(torch.randint(1, 5 * n + 2, (1,), generator=gen))
        g = random_graph(gen, n, e, 0, hub=70 if i == 1 and e > 80 else None)
        g['v'] = torch.rand(n, C, generator=gen) * 2 - 1
        g['goal'] = g['v'][0].clone()
        g['obstacles'] = torch.rand(int(torch.randint(0, 9, (1,), generator=gen)), S, generator=gen) - 0.5
        graphs.append(g)
    loop = int(torch.randint(1, 5, (1,), generator=gen))
    b = gnnmp.GraphBatch.from_graphs(graphs, S, DEV)
    exact = [x.cpu() for x in b.split_edges(m.forward_batch(b, loop))]
    if mode.endswith('fp32'):
        for g, part in zip(graphs, exact):
            check(part, w, g, loop)
        return
    m.mlp_dtype = 'bf16x3' if mode.endswith('bf16x3') else 'bf16'
    other = [x.cpu() for x in b.split_edges(m.forward_batch(b, loop))]
    for g, part, ref in zip(graphs, other, exact):
        if mode.endswith('bf16x3'):
            check(part, w, g, loop)
        else:
            err = (part - ref).abs()
            assert err.mean().item() <= 0.03 and err.max().item() <= 0.25, (err.mean().item(), err.max().item())


@pytest.mark.parametrize('tiles', [376, 384, 392])
def test_message_kernel_form_boundary(tiles):
    """The message kernel switches from one tile per workgroup to one tile per wave above 384 32-node tiles (d = 32): batches
    just below, at and above the switch give the same bits as their graphs scored one by one, and match the oracle."""
    gen = torch.Generator().manual_seed(tiles)
    w = load_weights('weights_maze')
    m = gnnmp.EncoderProcessDecoder(2, 2, 32, 2).eval()
    m.load_state_dict(w)
    graphs, have = [], 0
    while have < tiles:                                   # every graph occupies a multiple of 8 tiles (256-node padding)
        want = min(tiles - have, 8 * int(torch.randint(1, 4, (1,), generator=gen)))
        n = want * 32 - int(torch.randint(0, 200, (1,), generator=gen))
        graphs.append(random_graph(gen, n, int(torch.randint(n, 4 * n, (1,), generator=gen)), int(torch.randint(0, 60, (1,), generator=gen))))
        have += want
    b = gnnmp.GraphBatch.from_graphs(graphs, 2, DEV)
    parts = b.split_edges(m.forward_batch(b, 3))
    for i, (g, part) in enumerate(zip(graphs, parts)):
        one = m.edge_scores(g['goal'].to(DEV), 3, g['v'].to(DEV), g['obstacles'].to(DEV), g['edge_index'].to(DEV))
        assert torch.equal(part, one), i
    check(parts[0].cpu(), w, graphs[0], 3)
    check(parts[-1].cpu(), w, graphs[-1], 3)


def test_column_split_csr_build_ragged_batch():
    """Graphs averaging more than 8 k edges take the two-launch CSR build (edge columns of a graph split over several
    workgroups): a ragged batch with large graphs, an edge-less graph, a single-node graph and a hub; bit-equality with
    the graphs scored one by one (other part counts, or the one-launch build) and the oracle on the small ones."""
    gen = torch.Generator().manual_seed(909)
    w = load_weights('weights_maze')
    m = gnnmp.EncoderProcessDecoder(2, 2, 32, 2).eval()
    m.load_state_dict(w)
    graphs = [random_graph(gen, 2500, 40000, 30, hub=500), random_graph(gen, 40, 0, 5), random_graph(gen, 1, 3, 0),
              random_graph(gen, 3000, 52000, 90), random_graph(gen, 150, 600, 116, hub=90), random_graph(gen, 1800, 25000, 1)]
    b = gnnmp.GraphBatch.from_graphs(graphs, 2, DEV)
    assert b.total_edges // b.n_graphs > 2 * 8192
    parts = b.split_edges(m.forward_batch(b, 2))
    for i, (g, part) in enumerate(zip(graphs, parts)):
        if g['edge_index'].shape[1] == 0:
            assert part.numel() == 0
            continue
        one = m.edge_scores(g['goal'].to(DEV), 2, g['v'].to(DEV), g['obstacles'].to(DEV), g['edge_index'].to(DEV))
        assert torch.equal(part, one), i
    check(parts[2].cpu(), w, graphs[2], 2)
    check(parts[4].cpu(), w, graphs[4], 2)
    check(parts[5].cpu(), w, graphs[5], 2)

"""GPU parity of the HIP smoother forward against the goldens recorded from the reference and the
CPU oracle.  Tolerance: outputs are O(1) coordinates; reference fp32-vs-fp64 differs by <= 2e-6, the
bar is allclose(rtol=1e-5, atol=1e-5)."""
import os

import numpy as np
import pytest
import torch

from conftest import golden_files, load_weights
import gnnmp
from oracle import ref_cpu

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
CONF = {'smooth_2d_attv3': (2, 1.0), 'smooth_7d_attv3': (7, 1.0), 'smooth_ur5_attv3': (6, 2 * np.pi),
        'smooth_snake_attv3': (7, 1.0), 'smooth_13d_attv3': (13, 1.0), 'smooth_14d_attv3': (14, 1.0)}


def make(name):
    C, scale = CONF[name]
    m = gnnmp.ModelSmoother(workspace_size=3, config_size=C, embed_size=128, obs_size=6, scale=scale).eval()
    m.load_state_dict(load_weights(name), strict=True)
    return m


def chain_edges(P):
    a, b = torch.arange(1, P), torch.arange(0, P - 1)
    return torch.cat((torch.stack((a, b)), torch.stack((b, a)), torch.stack((torch.arange(P),) * 2)), dim=1)


@pytest.mark.parametrize('path', golden_files('smoother_'), ids=os.path.basename)
def test_golden(path):
    with np.load(path) as f:
        r = {k: f[k] for k in f.files}
    name = os.path.basename(path).split('_P')[0].replace('smoother_', '')
    m = make(name)
    p_in = torch.from_numpy(r['path']).to(DEV)
    keep = p_in.clone()
    out = m(path=p_in, free=torch.from_numpy(r['free']).to(DEV), collided=torch.from_numpy(r['collided']).to(DEV),
            obstacles=torch.zeros(1, 6, device=DEV), edge_index=torch.from_numpy(r['edge_index']).to(DEV),
            loop=int(r['loop'])).cpu()
    assert torch.equal(keep, p_in)                    # caller's path untouched
    ref32 = torch.from_numpy(r['out_fp32'])
    e32 = (out - ref32).abs().max().item()
    if 'out_fp64' in r:
        e64 = (out.double() - torch.from_numpy(r['out_fp64'])).abs().max().item()
        print('\n%s: max|gpu-ref32|=%.2e max|gpu-ref64|=%.2e' % (os.path.basename(path), e32, e64))
    else:
        # float32-kNN fixture (model_smoother.py:125 runs torch_cluster in the input dtype): the kernel's float32
        # distances + lower-index tie rule pick the neighbours of the float32 run, not those of the float64 run
        other = (out - torch.from_numpy(r['out_fp32_knn64'])).abs().max().item()
        print('\n%s: max|gpu-ref32(float32 kNN)|=%.2e, vs the float64-kNN run %.2e' % (os.path.basename(path), e32, other))
        assert other > 1e-4
    assert torch.allclose(out, ref32, rtol=1e-5, atol=1e-5), e32
    # end points are passed through (x / scale * scale), interior moved
    assert torch.equal(out[0], ref32[0]) and torch.equal(out[-1], ref32[-1])


@pytest.mark.parametrize('case', ['few_samples', 'no_collided_but_one', 'two_nodes', 'dup_edges', 'loop0'])
def test_edge_cases(case):
    gen = torch.Generator().manual_seed(11)
    name = 'smooth_2d_attv3'
    P, F, Co, loop = 9, 40, 30, 2
    if case == 'few_samples':
        F, Co = 4, 3                                 # fewer than k = 10 samples in total
    elif case == 'no_collided_but_one':
        F, Co = 25, 1                                # caller substitutes one zero row for an empty list (smoother.py:53-56)
    elif case == 'two_nodes':
        P = 2                                        # nothing to move: path[1:-1] is empty
    elif case == 'loop0':
        loop = 0
    path = torch.rand(P, 2, generator=gen) * 2 - 1
    free = torch.rand(F, 2, generator=gen) * 2 - 1
    coll = torch.rand(Co, 2, generator=gen) * 2 - 1 if case != 'no_collided_but_one' else torch.zeros(1, 2)
    ei = chain_edges(P)
    if case == 'dup_edges':
        ei = torch.cat((ei, ei[:, :5]), dim=1)       # coalesce must drop duplicates
    m = make(name)
    out = m(path=path.to(DEV), free=free.to(DEV), collided=coll.to(DEV), obstacles=None, edge_index=ei.to(DEV),
            loop=loop).cpu()
    ref = ref_cpu.smoother_forward(load_weights(name), path, free, coll, ei, loop=loop)
    assert torch.allclose(out, ref, rtol=1e-5, atol=1e-5), (out - ref).abs().max()


def test_batch_equals_single_bitwise():
    gen = torch.Generator().manual_seed(5)
    name = 'smooth_7d_attv3'
    m = make(name)
    probs = []
    for P, F, Co in [(5, 30, 20), (33, 100, 80), (12, 11, 1), (20, 500, 500)]:
        probs.append((torch.rand(P, 7, generator=gen) * 2 - 1, torch.rand(F, 7, generator=gen) * 2 - 1,
                      torch.rand(Co, 7, generator=gen) * 2 - 1, chain_edges(P)))
    sb = gnnmp.SmoothBatch([p[0] for p in probs], [p[1] for p in probs], [p[2] for p in probs], [p[3] for p in probs], DEV)
    out = m.forward_batch(sb, 1)
    off = 0
    w = load_weights(name)
    for path, free, coll, ei in probs:
        single = m(path=path.to(DEV), free=free.to(DEV), collided=coll.to(DEV), edge_index=ei.to(DEV), loop=1)
        assert torch.equal(single, out[off:off + path.shape[0]])
        ref = ref_cpu.smoother_forward(w, path, free, coll, ei, loop=1)
        assert torch.allclose(single.cpu(), ref, rtol=1e-5, atol=1e-5)
        off += path.shape[0]


@pytest.mark.parametrize('mode', ['fp32', 'bf16'])
def test_split_kernels_equal_wave_kernels_bitwise(mode):
    """The tile-per-workgroup kernels (a tile's layers split over four waves; all fp32 batches, bf16 up to 2048 edge
    tiles) and the tile-per-wave kernels (larger bf16 batches): same bits.  bf16: a > 2048-tile batch against single calls;
    fp32: the same comparison exercises batched vs single launches of the split kernels (the wave kernels are compared
    in tests/test_smoother_wave_kernels_gpu.py through the GNNMP_SM_SPLIT switch)."""
    gen = torch.Generator().manual_seed(17)
    name = 'smooth_7d_attv3'
    m = make(name)
    m.mlp_dtype = mode
    probs = []
    for i in range(240):                                 # ~9-14 edge tiles each: > 2048 tiles in the batch
        P = 20 + (i % 5) * 4
        probs.append((torch.rand(P, 7, generator=gen) * 2 - 1, torch.rand(150, 7, generator=gen) * 2 - 1,
                      torch.rand(100, 7, generator=gen) * 2 - 1, chain_edges(P)))
    sb = gnnmp.SmoothBatch([p[0] for p in probs], [p[1] for p in probs], [p[2] for p in probs], [p[3] for p in probs], DEV)
    assert sum((p[3].shape[1] + 10 * p[0].shape[0] + 31) // 32 for p in probs) > 2048
    out = m.forward_batch(sb, 2)
    off = 0
    for path, free, coll, ei in probs[:12]:
        single = m(path=path.to(DEV), free=free.to(DEV), collided=coll.to(DEV), edge_index=ei.to(DEV), loop=2)
        assert torch.equal(single, out[off:off + path.shape[0]])
        off += path.shape[0]


@pytest.mark.parametrize('seed', range(8))
def test_random_problems(seed):
    """Structure fuzz: random waypoint counts (incl. paths longer than one 32-row tile), sample counts from fewer than
    k = 10 up to the 1000-sample cap of the planner, random extra caller edges (into path nodes and into samples,
    duplicates, self loops), every checkpoint's dimension, scale and loop count, ragged batches."""
    gen = torch.Generator().manual_seed(900 + seed)
    name = list(CONF)[seed % len(CONF)]
    C, scale = CONF[name]
    m = make(name)
    w = load_weights(name)
    loop = int(torch.randint(1, 4, (1,), generator=gen))
    probs = []
    for _ in range(int(torch.randint(1, 5, (1,), generator=gen))):
        P = int(torch.randint(2, 70, (1,), generator=gen))
        F = int(torch.randint(1, 520, (1,), generator=gen))
        Co = int(torch.randint(1, 520, (1,), generator=gen))
        M = P + F + Co
        extra = torch.randint(0, M, (2, int(torch.randint(0, 40, (1,), generator=gen))), generator=gen)
        mk = lambda n: (torch.rand(n, C, generator=gen) * 2 - 1) * scale       # noqa: E731
        probs.append((mk(P), mk(F), mk(Co), torch.cat((chain_edges(P), extra), dim=1)))
    sb = gnnmp.SmoothBatch([p[0] for p in probs], [p[1] for p in probs], [p[2] for p in probs], [p[3] for p in probs], DEV)
    out = m.forward_batch(sb, loop).cpu()
    off = 0
    for path, free, coll, ei in probs:
        ref = ref_cpu.smoother_forward(w, path, free, coll, ei, loop=loop, scale=scale)
        got = out[off:off + path.shape[0]]
        assert torch.allclose(got, ref, rtol=1e-5, atol=1e-5 * scale), (name, (got - ref).abs().max())
        off += path.shape[0]


def test_limits_fail_loudly():
    """More samples than the kNN kernel holds per lane (2048) is an error, never a silent truncation."""
    m = make('smooth_2d_attv3')
    gen = torch.Generator().manual_seed(1)
    path, free, coll = (torch.rand(n, 2, generator=gen).to(DEV) for n in (6, 1500, 700))
    with pytest.raises(ValueError, match='at most 2048'):                 # the Python wrapper names the limit
        m(path=path, free=free, collided=coll, edge_index=chain_edges(6).to(DEV), loop=1)
    import ctypes
    from gnnmp import smoother as S
    sb = S.SmoothBatch([path], [free], [coll], [chain_edges(6).to(DEV)], DEV)
    cb = S._cbatch(sb)
    need = ctypes.c_size_t()
    L, h = S._lib.lib(), m._native(DEV)
    assert L.gnnmp_smoother_workspace_bytes(h, ctypes.byref(cb), ctypes.byref(need)) == 0
    ws = torch.empty(need.value, dtype=torch.uint8, device=DEV)
    out = torch.empty_like(sb.path)
    rc = L.gnnmp_smoother_forward(h, ctypes.byref(cb), 1, out.data_ptr(), ws.data_ptr(), ws.numel(), None)
    assert rc != 0 and b'dimensions' in L.gnnmp_status_string(rc)        # the C ABI refuses it as well (GNNMP_ERR_DIMS)
    long_path = torch.rand(800, 2, generator=gen).to(DEV)                # 3 * 800 chain edges + 10 * 800 kNN candidates
    with pytest.raises(ValueError, match='candidate edges'):
        m(path=long_path, free=free[:500], collided=coll[:500], edge_index=chain_edges(800).to(DEV), loop=1)

"""The fp32 parity bar shared by the GPU parity tests, smoke() and tools/parity_report.py.

north_star asks for "within 1e-5 fp32".  Edge scores span roughly [-32, +12] and the reference itself, run in fp32,
differs from the same module run in fp64 by `own` = max|ref_fp32 - ref_fp64| (3e-6 ... 2.4e-5 depending on the fixture;
profiles/r02_parity.txt lists every one), so the bar is per fixture:

    atol = max(1e-5, 1.25 * own)
    |gpu - ref_fp32| <= atol + 1e-5 * |ref_fp32|    elementwise        (bar32)
    |gpu - ref_fp64| <= atol + 1e-5 * |ref_fp64|    elementwise        (bar64: no worse than the reference's own fp32)

Fixtures whose `own` is below 8e-6 are therefore held to the bare north_star figure; the rest cannot meet a bare
1e-5 because the reference's own fp32 run does not."""
import torch

RTOL = 1e-5
ATOL_FLOOR = 1e-5
OWN_FACTOR = 1.25


def own_error(ref32, ref64):
    return (ref32.double() - ref64.double()).abs().max().item() if ref32.numel() else 0.0


def atol_for(own, own_factor=OWN_FACTOR):
    return max(ATOL_FLOOR, own_factor * own)


def check(gpu, ref32, ref64, own_factor=OWN_FACTOR):
    """Returns a dict with the three maxima, the bar and pass flags."""
    gpu, ref32, ref64 = gpu.double().cpu(), ref32.double().cpu(), ref64.double().cpu()
    own = own_error(ref32, ref64)
    atol = atol_for(own, own_factor)
    d32, d64 = (gpu - ref32).abs(), (gpu - ref64).abs()
    ok32 = bool((d32 <= atol + RTOL * ref32.abs()).all())
    ok64 = bool((d64 <= atol + RTOL * ref64.abs()).all())
    return dict(err32=d32.max().item() if d32.numel() else 0.0, err64=d64.max().item() if d64.numel() else 0.0, own=own,
                atol=atol, ok32=ok32, ok64=ok64, bare_1e5=bool((d32 <= 1e-5).all()), n_over_1e5=int((d32 > 1e-5).sum()),
                max_abs_ref=ref32.abs().max().item() if ref32.numel() else 0.0, n=int(gpu.numel()))


def assert_fp32_parity(gpu, ref32, ref64, what='', own_factor=OWN_FACTOR):
    """own_factor: 1.25 for the goldens / full-size / seeded-oracle checks; the structure-fuzz tests pass 2.5 because
    the max over a few hundred elements of two fp32 summation orders is a noisy statistic on tiny inputs (an indexing
    or segmentation bug shows as 1e-2 or more)."""
    r = check(gpu, ref32, ref64, own_factor)
    assert r['ok32'], '%s: max|gpu-ref32| %.3e over the bar %.3e + 1e-5|ref| (reference fp32-vs-fp64 %.3e)' % (
        what, r['err32'], r['atol'], r['own'])
    assert r['ok64'], '%s: max|gpu-ref64| %.3e over the bar %.3e + 1e-5|ref| (reference fp32-vs-fp64 %.3e)' % (
        what, r['err64'], r['atol'], r['own'])
    return r


def oracle_pair(fwd, weights, *tensors, **kw):
    """(ref32, ref64) of an oracle forward `fwd(weights, *tensors, **kw)`: the same function run in fp32 and in fp64."""
    ref32 = fwd(weights, *tensors, **kw)
    w64 = {k: (t.double() if t.is_floating_point() else t) for k, t in weights.items()}
    t64 = [(t.double() if torch.is_tensor(t) and t.is_floating_point() else t) for t in tensors]
    return ref32, fwd(w64, *t64, **kw)


def explorer_oracle_pair(w, g, loop, **kw):
    """fp32 and fp64 runs of the explorer oracle on graph dict g (v, goal, obstacles, edge_index)."""
    from oracle import ref_cpu
    r32 = ref_cpu.explorer_forward(w, g['v'], g['goal'], g['obstacles'], g['edge_index'], loop, **kw)
    w64 = {k: (t.double() if t.is_floating_point() else t) for k, t in w.items()}
    r64 = ref_cpu.explorer_forward(w64, g['v'].double(), g['goal'].double(), g['obstacles'].double(), g['edge_index'], loop, **kw)
    return r32, r64

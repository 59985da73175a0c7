"""The fp32 parity bar shared by the GPU parity tests, smoke() and tools/parity_report.py.

north_star asks for "within 1e-5 fp32".  Edge scores span roughly [-32, +12] and the reference itself, run in fp32,
differs from the same module run in fp64 by `own` = max|ref_fp32 - ref_fp64| (3e-6 ... 2.4e-5 depending on the fixture;
profiles/r03_parity.txt lists every one).  The bar is ABSOLUTE (no relative term) and per fixture:

    bar64:  |gpu - ref_fp64| <= max(1e-5, 1.25 * own)          elementwise   (no further from the exact result than the
                                                                              reference's own fp32 run, 25 % margin)
    bar32:  |gpu - ref_fp32| <= max(1e-5, 1.25 * own) + own    elementwise   (ref_fp32 is itself `own` away from the exact
                                                                              result: triangle inequality on bar64)
    bare:   |gpu - ref_fp32| <= 1e-5 wherever own <= 1e-5      (tests/test_explorer_parity.py::test_golden_bare_1e5;
                                                                the north_star figure, asserted on every golden fixture on
                                                                which the reference's own fp32 run holds it)

Round 3 measured max|gpu - ref_fp64| <= 9.3e-6 on every explorer golden (the bare figure against the EXACT result holds
everywhere) since the node side's block 0 runs in double precision (csrc/explorer_kernels.hip node_f64_body)."""
import torch

ATOL_FLOOR = 1e-5
OWN_FACTOR = 1.25


def own_error(ref32, ref64):
    return (ref32.double() - ref64.double()).abs().max().item() if ref32.numel() else 0.0


def atol_for(own, own_factor=OWN_FACTOR):
    return max(ATOL_FLOOR, own_factor * own)


def check(gpu, ref32, ref64, own_factor=OWN_FACTOR):
    """Returns a dict with the three maxima, the bars and pass flags."""
    gpu, ref32, ref64 = gpu.double().cpu(), ref32.double().cpu(), ref64.double().cpu()
    own = own_error(ref32, ref64)
    atol = atol_for(own, own_factor)
    d32, d64 = (gpu - ref32).abs(), (gpu - ref64).abs()
    ok32 = bool((d32 <= atol + own).all())
    ok64 = bool((d64 <= atol).all())
    strict = d32 > (1e-5 + 1e-5 * ref32.abs())
    return dict(err32=d32.max().item() if d32.numel() else 0.0, err64=d64.max().item() if d64.numel() else 0.0, own=own,
                atol=atol, atol32=atol + own, ok32=ok32, ok64=ok64, bare_1e5=bool((d32 <= 1e-5).all()),
                n_over_1e5=int((d32 > 1e-5).sum()), n_over_1e5_64=int((d64 > 1e-5).sum()), n_fail_allclose=int(strict.sum()),
                max_abs_ref=ref32.abs().max().item() if ref32.numel() else 0.0, n=int(gpu.numel()))


def assert_fp32_parity(gpu, ref32, ref64, what='', own_factor=OWN_FACTOR):
    r = check(gpu, ref32, ref64, own_factor)
    assert r['ok64'], '%s: max|gpu-ref64| %.3e over the bar %.3e (reference fp32-vs-fp64 %.3e)' % (
        what, r['err64'], r['atol'], r['own'])
    assert r['ok32'], '%s: max|gpu-ref32| %.3e over the bar %.3e + %.3e (reference fp32-vs-fp64)' % (
        what, r['err32'], r['atol'], r['own'])
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

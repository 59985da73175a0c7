"""CPU check of the MFMA operand packing (gnnmp_pack_* in include/gnnmp.h) against a numpy
emulation of v_mfma_f32_32x32x2_f32's lane/register layout: a chain of packed layers evaluated
exactly the way the kernels' linear_acc / linear_in do must equal plain X @ W^T."""

import numpy as np
import pytest

import gnnmp  # noqa: F401
from gnnmp import _lib


def phi(r, h):
    return (r & 3) + 8 * (r >> 2) + 4 * h


LANE = np.arange(64)


def mfma(a, b, acc):
    """acc[r][lane] += sum_k A[i][k] B[k][j]; A[i=l&31][k=l>>5]=a[l]; B[k=l>>5][j=l&31]=b[l];
    D lane l, reg r <-> (i = phi(r, l>>5), j = l&31)."""
    A = np.zeros((32, 2)); B = np.zeros((2, 32))
    A[LANE & 31, LANE >> 5] = a
    B[LANE >> 5, LANE & 31] = b
    Dm = A @ B
    for r in range(16):
        acc[r] += Dm[phi(r, LANE >> 5), LANE & 31]
    return acc


def to_regs(X, nt):
    """X [32 rows, 32*nt] -> x[t][r][lane]."""
    return np.stack([np.stack([X[LANE & 31, 32 * t + phi(r, LANE >> 5)] for r in range(16)]) for t in range(nt)])


def from_regs(x):
    nt = x.shape[0]
    X = np.zeros((32, 32 * nt))
    for t in range(nt):
        for r in range(16):
            X[LANE & 31, 32 * t + phi(r, LANE >> 5)] = x[t][r]
    return X


def pack_tiles(W, col0, n_in):
    out_f, ld = W.shape
    dst = np.zeros((out_f // 32) * (n_in // 32) * 1024, dtype=np.float32)
    Wc = np.ascontiguousarray(W, dtype=np.float32)
    n = _lib.lib().gnnmp_pack_a_tiles(Wc.ctypes.data, out_f, ld, col0, n_in, dst.ctypes.data)
    assert n == dst.size
    return dst


def linear_acc(A, x, y, nto, nti):
    for it in range(nti):
        for ot in range(nto):
            tile = A[(ot * nti + it) * 1024:(ot * nti + it + 1) * 1024]
            for q in range(4):
                w = tile[(q * 64 + LANE[:, None]) * 4 + np.arange(4)[None, :]]      # [lane, 4]
                for c in range(4):
                    y[ot] = mfma(w[:, c], x[it][q * 4 + c], y[ot])
    return y


@pytest.mark.parametrize('d', [32, 64])
def test_tile_chain_matches_matmul(d):
    rng = np.random.default_rng(d)
    nt = d // 32
    X = rng.standard_normal((32, d))
    W1 = rng.standard_normal((d, 3 * d)).astype(np.float32)     # use the middle column block
    W2 = rng.standard_normal((d, d)).astype(np.float32)
    A1 = pack_tiles(W1, d, d)
    A2 = pack_tiles(W2, 0, d)
    x = to_regs(X, nt)
    y = linear_acc(A1, x, np.zeros((nt, 16, 64)), nt, nt)
    y = np.maximum(y, 0)
    z = linear_acc(A2, y, np.zeros((nt, 16, 64)), nt, nt)
    ref = np.maximum(X @ W1[:, d:2 * d].T.astype(np.float64), 0) @ W2.T.astype(np.float64)
    assert np.allclose(from_regs(z), ref, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize('n_in', [2, 5, 8, 28])
def test_small_first_layer(n_in):
    d = 64
    nt = d // 32
    rng = np.random.default_rng(n_in)
    W = rng.standard_normal((d, n_in)).astype(np.float32)
    X = rng.standard_normal((32, n_in))
    ks = (n_in + 1) // 2
    dst = np.zeros(nt * ks * 64, dtype=np.float32)
    assert _lib.lib().gnnmp_pack_a_small(W.ctypes.data, d, n_in, 0, n_in, dst.ctypes.data) == dst.size
    y = np.zeros((nt, 16, 64))
    for st in range(ks):
        k = 2 * st + (LANE >> 5)
        b = np.where(k < n_in, X[LANE & 31, np.minimum(k, n_in - 1)], 0.0)
        for ot in range(nt):
            y[ot] = mfma(dst[(ot * ks + st) * 64 + LANE], b, y[ot])
    assert np.allclose(from_regs(y), X @ W.T.astype(np.float64), rtol=1e-6, atol=1e-6)


def test_vec_order():
    d = 64
    b = np.arange(d, dtype=np.float32)
    dst = np.zeros(d, dtype=np.float32)
    assert _lib.lib().gnnmp_pack_vec(b.ctypes.data, d, dst.ctypes.data) == d
    for t in range(2):
        for h in range(2):
            for r in range(16):
                assert dst[(t * 2 + h) * 16 + r] == 32 * t + phi(r, h)


def test_obstacle_value_transpose_rule():
    """obs_kernel writes V[o][f] (held as row j = o, register r, half h) into the A tile of the
    P.V product, whose (lane', r') element must be V[o = phi(r', lane'>>5)][f = lane'&31]."""
    for j in range(32):
        hp, rp = (j >> 2) & 1, (j & 3) + 4 * (j >> 3)
        assert phi(rp, hp) == j


# ---- v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32 operands of the double-precision node stretch (gnnmp_pack_f64_ops).
# Layouts probed on the MI355X with tools/microbench/mfma_f64_probe.hip: A[i][k] in lane i + 16 k, B[k][j] in lane j + 16 k;
# D[i][j] of the f64 instruction in lane j + 16 (i % 4), register i / 4; of the f32 one in lane j + 16 (i / 4), register i % 4.
def mfma16(a, b, acc, f64):
    A = np.zeros((16, 4)); B = np.zeros((4, 16))
    A[LANE & 15, LANE >> 4] = a
    B[LANE >> 4, LANE & 15] = b
    Dm = A @ B
    for r in range(4):
        i = 4 * r + (LANE >> 4) if f64 else 4 * (LANE >> 4) + r
        acc[r] += Dm[i, LANE & 15]
    return acc


def pack_f64(W, n_in, row_perm):
    out_f, ld = W.shape
    nks = (n_in + 3) // 4
    dst = np.zeros((out_f // 16) * nks * 64, dtype=np.float32)
    Wc = np.ascontiguousarray(W, dtype=np.float32)
    n = _lib.lib().gnnmp_pack_f64_ops(Wc.ctypes.data, out_f, ld, 0, n_in, row_perm, dst.ctypes.data)
    assert n == dst.size
    return dst.reshape(out_f // 16, nks, 64)


@pytest.mark.parametrize('D,C', [(32, 2), (64, 7), (32, 14)])
def test_f64_chain_layout(D, C):
    """node_f64_body's chain: first layer on C raw inputs, a d x d layer on the f64 instruction, and a d x d layer on the f32
    instruction with row-permuted operands landing in the f64 register layout -- all equal to plain matrix products."""
    rng = np.random.default_rng(D + C)
    NB = D // 16
    X = rng.standard_normal((16, C)).astype(np.float32)                 # 16 nodes
    W1 = rng.standard_normal((D, C)).astype(np.float32)
    W2 = rng.standard_normal((D, D)).astype(np.float32)
    W3 = rng.standard_normal((D, D)).astype(np.float32)
    j, g = LANE & 15, LANE >> 4
    A1 = pack_f64(W1, C, 0)
    h = np.zeros((NB, 4, 64))
    for st in range(A1.shape[1]):
        k = 4 * st + g
        xin = np.where(k < C, X[j, np.minimum(k, C - 1)], 0.0)
        for ob in range(NB):
            mfma16(A1[ob, st].astype(np.float64), xin, h[ob], True)
    ref1 = X.astype(np.float64) @ W1.T.astype(np.float64)
    for ob in range(NB):
        for r in range(4):
            assert np.allclose(h[ob, r], ref1[j, 16 * ob + 4 * r + g], rtol=1e-12, atol=1e-12)
    for perm, f64 in ((0, True), (1, False)):
        A = pack_f64(W2 if f64 else W3, D, perm).reshape(NB, NB, 4, 64)
        y = np.zeros((NB, 4, 64))
        for ib in range(NB):
            for st in range(4):
                for ob in range(NB):
                    mfma16(A[ob, ib, st].astype(np.float64), h[ib, st], y[ob], f64)
        ref = ref1 @ (W2 if f64 else W3).T.astype(np.float64)
        for ob in range(NB):
            for r in range(4):                                         # both land in the f64 layout: feature 16 ob + 4 r + g
                assert np.allclose(y[ob, r], ref[j, 16 * ob + 4 * r + g], rtol=1e-12, atol=1e-9)


def test_f64_pack_rejects_bad_arguments():
    w = np.zeros((16, 4), dtype=np.float32)
    dst = np.zeros(16 * 64, dtype=np.float32)
    L = _lib.lib()
    assert L.gnnmp_pack_f64_ops(w.ctypes.data, 15, 4, 0, 4, 0, dst.ctypes.data) < 0        # rows not a multiple of 16
    assert L.gnnmp_pack_f64_ops(w.ctypes.data, 16, 4, 0, 0, 0, dst.ctypes.data) < 0        # no input columns
    assert L.gnnmp_pack_f64_ops(None, 16, 4, 0, 4, 0, dst.ctypes.data) < 0
    assert L.gnnmp_pack_f64_ops(w.ctypes.data, 16, 4, 0, 4, 0, dst.ctypes.data) == 64      # one 16-row block, one k-step

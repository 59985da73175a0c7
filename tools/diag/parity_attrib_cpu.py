#!/usr/bin/env python
"""CPU attribution experiment (round 3): which of the attention kernel's re-formulations moves fp32 results away from
the fp64 run?  The oracle's `_attention` is swapped for an fp32 emulation of the KERNEL's formulation (k-ordered fmaf
chains like the fp32 MFMA, Wqk fold, exp2 with folded scale, online softmax over 32-obstacle tiles, reciprocal) with one
switch per approximation; everything else stays in the reference formulation.  Prints max|variant32 - ref64| per fixture.
Runs anywhere (no GPU)."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from oracle import ref_cpu  # noqa: E402
from gnnmp.synth import ENVS  # noqa: E402
from gnnmp.weights import load_weights  # noqa: E402

f32 = np.float32


def chain_mm(A, B):
    """C[i, j] = k-ordered fmaf chain of A[i, k] * B[k, j] in fp32 (product exact in double, one rounding per step)."""
    A = A.astype(np.float64); B = B.astype(np.float64)
    acc = np.zeros((A.shape[0], B.shape[1]), f32)
    for k in range(A.shape[1]):
        acc = (acc.astype(np.float64) + A[:, k:k + 1] * B[k:k + 1, :]).astype(f32)
    return acc


def chain_mm_acc(acc, A, B):
    A = A.astype(np.float64); B = B.astype(np.float64)
    for k in range(A.shape[1]):
        acc = (acc.astype(np.float64) + A[:, k:k + 1] * B[k:k + 1, :]).astype(f32)
    return acc


def make_attention(sw):
    def att(w, pre, m_t, o_t, materialize=False):
        if m_t.dtype != torch.float32:
            return ORIG(w, pre, m_t, o_t, materialize)
        m = m_t.numpy(); o = o_t.numpy()
        d = m.shape[1]
        Wq = w[pre + '.query.weight'].numpy(); Wk = w[pre + '.key.weight'].numpy(); Wv = w[pre + '.value.weight'].numpy()
        mv = chain_mm(m, Wv.T)
        ov = chain_mm(o, Wv.T)
        if sw['fold']:
            Wqk = (Wq.astype(np.float64).T @ Wk.astype(np.float64)).astype(f32)        # [i][j] = sum_c Wq[c][i] Wk[c][j]
            kp = chain_mm(o, Wqk.T)                     # K'[o][i] = sum_j Wqk[i][j] code[o][j]
            tq = chain_mm(m, Wqk.T)
            obs = chain_mm(m, kp.T)
            if sw.get('self_unfused'):
                mq = chain_mm(m, Wq.T); mk = chain_mm(m, Wk.T)
                self_ = np.zeros(len(m), f32)
                for k in range(d):
                    self_ = (self_.astype(np.float64) + mq[:, k].astype(np.float64) * mk[:, k]).astype(f32)
            else:
                self_ = np.zeros(len(m), f32)
                for k in range(d):
                    self_ = (self_.astype(np.float64) + m[:, k].astype(np.float64) * tq[:, k]).astype(f32)
        else:
            mq = chain_mm(m, Wq.T); mk = chain_mm(m, Wk.T); ok = chain_mm(o, Wk.T)
            obs = chain_mm(mq, ok.T)
            self_ = np.zeros(len(m), f32)
            for k in range(d):
                self_ = (self_.astype(np.float64) + mq[:, k].astype(np.float64) * mk[:, k]).astype(f32)
        O = o.shape[0]
        if sw['online']:
            cs = f32(1.4426950408889634) / f32(np.sqrt(f32(d)))
            mx = self_.copy(); psum = np.ones(len(m), f32); acc = mv.copy()
            for o0 in range(0, O, 32):
                s = obs[:, o0:o0 + 32]
                nmx = np.maximum(mx, s.max(axis=1))
                if sw['exp2']:
                    off = (-nmx * cs).astype(f32)
                    arg = (s.astype(np.float64) * cs + off[:, None]).astype(f32)
                    p = np.exp2(arg.astype(np.float64)).astype(f32)
                    alpha = np.exp2(((mx - nmx).astype(f32) * cs).astype(np.float64)).astype(f32)
                else:
                    sq = f32(np.sqrt(f32(d)))
                    arg = ((s / sq).astype(f32) - (nmx / sq).astype(f32)[:, None]).astype(f32)
                    p = np.exp(arg.astype(np.float64)).astype(f32)
                    alpha = np.exp(((mx / sq).astype(f32) - (nmx / sq).astype(f32)).astype(np.float64)).astype(f32)
                # tree sum of p (pairs), then cross-half: emulate as pairwise
                ps = p.astype(f32)
                while ps.shape[1] > 1:
                    if ps.shape[1] % 2: ps = np.concatenate([ps, np.zeros((len(m), 1), f32)], 1)
                    ps = (ps[:, 0::2] + ps[:, 1::2]).astype(f32)
                psum = (psum * alpha).astype(f32)
                acc = (acc * alpha[:, None]).astype(f32)
                psum = (psum + ps[:, 0]).astype(f32)
                acc = chain_mm_acc(acc, p, ov[o0:o0 + 32])
                mx = nmx
            den = psum
        else:
            sq = f32(np.sqrt(f32(d)))
            allx = np.concatenate([self_[:, None], obs], 1)
            if sw['exp2']:
                cs = f32(1.4426950408889634) / sq
                mxx = allx.max(axis=1)
                off = (-mxx * cs).astype(f32)
                p = np.exp2((allx.astype(np.float64) * cs + off[:, None]).astype(f32).astype(np.float64)).astype(f32)
            else:
                xs = (allx / sq).astype(f32)
                p = np.exp((xs - xs.max(axis=1, keepdims=True)).astype(f32).astype(np.float64)).astype(f32)
            if sw.get('pairwise_pv'):
                den = p.sum(axis=1, dtype=np.float64).astype(f32)
                acc = (p.astype(np.float64) @ np.concatenate([np.zeros((1, d)), ov.astype(np.float64)], 0)
                       + p[:, :1].astype(np.float64) * mv).astype(f32)     # ~exact accumulation, rounded once
            else:
                den = np.zeros(len(m), f32)
                for k in range(p.shape[1]):
                    den = (den + p[:, k]).astype(f32)
                acc = (p[:, :1] * mv).astype(f32)
                acc = chain_mm_acc(acc, p[:, 1:], ov)
        if sw['rcp']:
            inv = (f32(1) / den).astype(f32)
            new = (acc.astype(np.float64) * inv[:, None] + m).astype(f32)       # fma
        else:
            new = ((acc / den[:, None]).astype(f32) + m).astype(f32)
        return ref_cpu._layer_norm(w, pre + '.layer_norm', torch.from_numpy(new), 1e-6)
    return att


ORIG = ref_cpu._attention
VARIANTS = {
    'kernel (fold,exp2,online,rcp)': dict(fold=1, exp2=1, online=1, rcp=1),
    'no fold': dict(fold=0, exp2=1, online=1, rcp=1),
    'fold, self unfused': dict(fold=1, exp2=1, online=1, rcp=1, self_unfused=1),
    'no exp2': dict(fold=1, exp2=0, online=1, rcp=1),
    'single pass': dict(fold=1, exp2=1, online=0, rcp=1),
    'no rcp': dict(fold=1, exp2=1, online=1, rcp=0),
    'none (chain mm only)': dict(fold=0, exp2=0, online=0, rcp=0),
    'none + exact PV': dict(fold=0, exp2=0, online=0, rcp=0, pairwise_pv=1),
    'fold + exact PV single pass': dict(fold=1, exp2=0, online=0, rcp=0, pairwise_pv=1),
}
fixtures = sys.argv[1:] or ['explorer_maze2_N64_k4_L5', 'explorer_maze2_N64_k4_L3', 'explorer_maze2_N64_k4_L1', 'explorer_maze2_N200_k6_L5', 'explorer_ur5_N64_k4_L5']
print('%-36s' % 'variant' + ''.join('%28s' % f.replace('explorer_', '') for f in fixtures))
rows = {}
for f in fixtures:
    with np.load(os.path.join(REPO, 'tests', 'golden', f + '.npz')) as z:
        r = {k: z[k] for k in z.files}
    env = f.split('_')[1]
    w = load_weights(ENVS[env]['ckpt'])
    args = [torch.from_numpy(r[k]) for k in ('v', 'goal', 'obstacles', 'edge_index')]
    ref64 = torch.from_numpy(r['scores_fp64']); ref32 = torch.from_numpy(r['scores_fp32'])
    rows.setdefault('reference fp32 (golden)', []).append((ref32.double() - ref64).abs().max().item())
    ref_cpu._attention = ORIG
    s = ref_cpu.explorer_forward(w, *args, int(r['loop']))
    rows.setdefault('oracle fp32', []).append((s.double() - ref64).abs().max().item())
    for name, sw in VARIANTS.items():
        ref_cpu._attention = make_attention(sw)
        s = ref_cpu.explorer_forward(w, *args, int(r['loop']))
        rows.setdefault(name, []).append((s.double() - ref64).abs().max().item())
    ref_cpu._attention = ORIG
for name, v in rows.items():
    print('%-36s' % name + ''.join('%28.3e' % x for x in v))

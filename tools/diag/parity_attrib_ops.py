#!/usr/bin/env python
"""Second CPU attribution experiment: the attention block in the REFERENCE formulation, fp32, with each matrix product /
sum accumulated either as a k-ordered fp32 fmaf chain (what v_mfma_f32_32x32x2_f32 does) or exactly (fp64 accumulate,
one rounding).  Which accumulation moves the scores away from the fp64 run?"""
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


def mm(A, B, mode, acc0=None):
    if mode == 'exact':
        r = A.astype(np.float64) @ B.astype(np.float64)
        if acc0 is not None: r = r + acc0
        return r.astype(f32)
    if mode == 'torch':
        r = (torch.from_numpy(np.ascontiguousarray(A)) @ torch.from_numpy(np.ascontiguousarray(B))).numpy()
        if acc0 is not None: r = (r + acc0).astype(f32)
        return r
    A = A.astype(np.float64); B = B.astype(np.float64)
    acc = np.zeros((A.shape[0], B.shape[1]), f32) if acc0 is None else acc0.astype(f32)
    if mode == 'chain':
        for k in range(A.shape[1]):
            acc = (acc.astype(np.float64) + A[:, k:k + 1] * B[k:k + 1, :]).astype(f32)
        return acc
    if mode.startswith('blk'):     # blocks of n k-steps, each a chain from zero, block results added in order
        n = int(mode[3:])
        for k0 in range(0, A.shape[1], n):
            part = np.zeros_like(acc)
            for k in range(k0, min(k0 + n, A.shape[1])):
                part = (part.astype(np.float64) + A[:, k:k + 1] * B[k:k + 1, :]).astype(f32)
            acc = (acc + part).astype(f32)
        return acc
    raise ValueError(mode)


def make_attention(md):
    def att(w, pre, m_t, o_t, materialize=False):
        if m_t.dtype != torch.float32:
            return ORIG(w, pre, m_t, o_t, materialize)
        m = m_t.numpy(); o = o_t.numpy()
        d = m.shape[1]
        Wq = w[pre + '.query.weight'].numpy(); Wk = w[pre + '.key.weight'].numpy(); Wv = w[pre + '.value.weight'].numpy()
        mv = mm(m, Wv.T, md['proj']); ov = mm(o, Wv.T, md['proj'])
        mq = mm(m, Wq.T, md['proj']); mk = mm(m, Wk.T, md['proj']); ok = mm(o, Wk.T, md['proj'])
        obs = mm(mq, ok.T, md['logit'])
        self_ = np.stack([mm(mq[i:i + 1], mk[i:i + 1].T, md['logit'])[0, 0] for i in range(len(m))]).astype(f32)
        sq = f32(np.sqrt(f32(d)))
        xs = (np.concatenate([self_[:, None], obs], 1) / sq).astype(f32)
        p = np.exp((xs - xs.max(axis=1, keepdims=True)).astype(f32).astype(np.float64)).astype(f32)
        den = mm(p, np.ones((p.shape[1], 1), f32), md['den'])[:, 0]
        if md.get('norm_first'):       # reference: softmax weights normalised, then the weighted sum
            pn = (p / den[:, None]).astype(f32)
            acc = (pn[:, :1] * mv).astype(f32)
            new = mm(pn[:, 1:], ov, md['pv'], acc)
            new = (new + m).astype(f32)
        else:
            acc = (p[:, :1] * mv).astype(f32)
            acc = mm(p[:, 1:], ov, md['pv'], acc)
            new = ((acc / den[:, None]).astype(f32) + m).astype(f32)
        return ref_cpu._layer_norm(w, pre + '.layer_norm', torch.from_numpy(new), 1e-6)
    return att


ORIG = ref_cpu._attention
C = 'chain'; E = 'exact'; T = 'torch'
VARIANTS = {
    'all chain': dict(proj=C, logit=C, den=C, pv=C),
    'all chain, normalise first': dict(proj=C, logit=C, den=C, pv=C, norm_first=1),
    'all torch matmul': dict(proj=T, logit=T, den=T, pv=T, norm_first=1),
    'all exact': dict(proj=E, logit=E, den=E, pv=E),
    'exact proj': dict(proj=E, logit=C, den=C, pv=C),
    'exact logit': dict(proj=C, logit=E, den=C, pv=C),
    'exact den': dict(proj=C, logit=C, den=E, pv=C),
    'exact pv': dict(proj=C, logit=C, den=C, pv=E),
    'exact den+pv': dict(proj=C, logit=C, den=E, pv=E),
    'blk32 den+pv': dict(proj=C, logit=C, den='blk32', pv='blk32'),
    'blk8 den+pv': dict(proj=C, logit=C, den='blk8', pv='blk8'),
    'blk8 everything': dict(proj='blk8', logit='blk8', den='blk8', pv='blk8'),
    'blk4 everything': dict(proj='blk4', logit='blk4', den='blk4', pv='blk4'),
    'exact proj+logit': dict(proj=E, logit=E, den=C, pv=C),
}
fixtures = sys.argv[1:] or ['explorer_maze2_N64_k4_L5', 'explorer_maze2_N64_k4_L3', 'explorer_maze2_N64_k4_L1', 'explorer_maze2_N200_k6_L5', 'explorer_ur5_N64_k4_L5']
print('%-36s' % 'variant' + ''.join('%22s' % f.replace('explorer_', '') for f in fixtures))
rows = {}
for f in fixtures:
    with np.load(os.path.join(REPO, 'tests', 'golden', f + '.npz')) as z:
        r = {k: z[k] for k in z.files}
    env = f.split('_')[1]
    w = load_weights(ENVS[env]['ckpt'])
    args = [torch.from_numpy(r[k]) for k in ('v', 'goal', 'obstacles', 'edge_index')]
    ref64 = torch.from_numpy(r['scores_fp64']); ref32 = torch.from_numpy(r['scores_fp32'])
    rows.setdefault('reference fp32 (golden)', []).append((ref32.double() - ref64).abs().max().item())
    for name, md in VARIANTS.items():
        ref_cpu._attention = make_attention(md)
        s = ref_cpu.explorer_forward(w, *args, int(r['loop']))
        rows.setdefault(name, []).append((s.double() - ref64).abs().max().item())
    ref_cpu._attention = ORIG
for name, v in rows.items():
    print('%-36s' % name + ''.join('%22.3e' % x for x in v))

# ---- distribution view: is the maximum one sensitive element?
print()
f = 'explorer_maze2_N64_k4_L5'
with np.load(os.path.join(REPO, 'tests', 'golden', f + '.npz')) as z:
    r = {k: z[k] for k in z.files}
w = load_weights(ENVS['maze2']['ckpt'])
args = [torch.from_numpy(r[k]) for k in ('v', 'goal', 'obstacles', 'edge_index')]
ref64 = torch.from_numpy(r['scores_fp64']); ref32 = torch.from_numpy(r['scores_fp32'])
def stats(name, s):
    e = (s.double() - ref64).abs()
    top = torch.topk(e, 3)
    print('%-30s rms %.3e  p90 %.3e  max %.3e  top idx %s  |ref| there %s' % (name, e.pow(2).mean().sqrt(), e.quantile(0.9), e.max(),
          top.indices.tolist(), [round(x, 2) for x in ref64[top.indices].tolist()]))
stats('reference fp32', ref32)
for name in ('all chain', 'all torch matmul', 'all exact', 'blk8 everything', 'blk4 everything', 'exact den+pv'):
    ref_cpu._attention = make_attention(VARIANTS[name])
    stats(name, ref_cpu.explorer_forward(w, *args, int(r['loop'])))
ref_cpu._attention = ORIG

#!/usr/bin/env python
"""Emulation: kernel formulation (k-ordered fp32 chains, Wqk fold, exp2, online softmax, rcp) everywhere, EXCEPT the
node_free_code encoder and parts of node attention block 0 in fp64.  What does the score error vs the fp64 run become?"""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import ref_cpu
from gnnmp.synth import ENVS
from gnnmp.weights import load_weights
import importlib
import io, contextlib
with contextlib.redirect_stdout(io.StringIO()):
    sys.argv = [sys.argv[0], 'explorer_ur5_N64_k4_L5']
    pa = importlib.import_module('parity_attrib_cpu')
KERNEL = dict(fold=1, exp2=1, online=1, rcp=1)
kern_att = pa.make_attention(KERNEL)
ORIG_ATT = pa.ORIG; ORIG_MLP2 = ref_cpu._mlp2
f32 = np.float32

def att_mixed(level):
    """level: 'full' = the whole attention sub-block of node block 0 in fp64 (result rounded to fp32);
    'post' = logits from the fp32 kernel formulation, softmax + PV + residual + LN in fp64."""
    def att(w, pre, m, o, materialize=False):
        if m.dtype != torch.float32 or not pre.startswith('node_attentions.0'):
            return kern_att(w, pre, m, o, materialize) if m.dtype == torch.float32 else ORIG_ATT(w, pre, m, o, materialize)
        w64 = {k: t.double() for k, t in w.items() if k.startswith(pre)}
        if level == 'full':
            return ORIG_ATT(w64, pre, m.double(), o.double()).float()
        d = m.shape[1]
        Wq = w[pre + '.query.weight'].numpy(); Wk = w[pre + '.key.weight'].numpy(); Wv = w[pre + '.value.weight'].numpy()
        mn, on = m.numpy(), o.numpy()
        Wqk = (Wq.astype(np.float64).T @ Wk.astype(np.float64)).astype(f32)
        kp = pa.chain_mm(on, Wqk.T); tq = pa.chain_mm(mn, Wqk.T)
        obs = pa.chain_mm(mn, kp.T).astype(np.float64)
        self_ = (mn.astype(np.float64) * tq).sum(1)      # (chain in the kernel; not critical)
        mv = pa.chain_mm(mn, Wv.T).astype(np.float64); ov = pa.chain_mm(on, Wv.T).astype(np.float64)
        a = torch.from_numpy(np.concatenate([self_[:, None], obs], 1)) / d ** 0.5
        p = a.softmax(-1).numpy()
        new = p[:, :1] * mv + p[:, 1:] @ ov
        x = torch.from_numpy(new) + m.double()
        return ref_cpu._layer_norm(w64, pre + '.layer_norm', x, 1e-6).float()
    return att

def mlp2_mixed(w_, name, x):
    if name == 'node_free_code' and x.dtype == torch.float32:
        w64 = {k: t.double() for k, t in w_.items() if k.startswith(name)}
        return ORIG_MLP2(w64, name, x.double()).float()
    return ORIG_MLP2(w_, name, x)

fixtures = ['explorer_maze2_N64_k4_L5', 'explorer_maze2_N64_k4_L3', 'explorer_maze2_N64_k4_L1', 'explorer_maze2_N200_k6_L5', 'explorer_ur5_N64_k4_L5', 'explorer_maze2_N1000_k8_L5']
if len(sys.argv) > 2: fixtures = sys.argv[2:]
print('%-44s' % 'variant' + ''.join('%20s' % f.replace('explorer_', '') for f in fixtures))
rows = {}
for f in fixtures:
    with np.load(os.path.join(REPO, 'tests', 'golden', f + '.npz')) as z:
        r = {k: z[k] for k in z.files}
    w = load_weights(ENVS[f.split('_')[1]]['ckpt'])
    args = [torch.from_numpy(r[k]) for k in ('v', 'goal', 'obstacles', 'edge_index')]
    ref64 = torch.from_numpy(r['scores_fp64']); ref32 = torch.from_numpy(r['scores_fp32'])
    def go(name, att, mlp2):
        ref_cpu._attention = att; ref_cpu._mlp2 = mlp2
        s = ref_cpu.explorer_forward(w, *args, int(r['loop']))
        rows.setdefault(name, []).append(((s.double() - ref64).abs().max().item(), (s.double() - ref32.double()).abs().max().item()))
    rows.setdefault('reference fp32', []).append(((ref32.double() - ref64).abs().max().item(), 0.0))
    go('kernel formulation', kern_att, ORIG_MLP2)
    go('+ nf encoder fp64', kern_att, mlp2_mixed)
    go('+ node blk0 softmax/PV/LN fp64', att_mixed('post'), ORIG_MLP2)
    go('+ both', att_mixed('post'), mlp2_mixed)
    go('+ nf enc + whole node blk0 attention fp64', att_mixed('full'), mlp2_mixed)
    ref_cpu._attention = ORIG_ATT; ref_cpu._mlp2 = ORIG_MLP2
for name, v in rows.items():
    print('%-44s' % name + ''.join('   %.2e/%.2e' % x for x in v))
print('(each cell: max|variant - ref64| / max|variant - ref32|)')

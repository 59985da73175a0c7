#!/usr/bin/env python
"""Finer stage sensitivity: inside node attention block 0 and per encoder, which op in fp32 (inside an fp64 run) makes the error."""
import os, sys
import numpy as np, torch
import torch.nn.functional as F
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from oracle import ref_cpu
from gnnmp.synth import ENVS
from gnnmp.weights import load_weights

f = sys.argv[1] if len(sys.argv) > 1 else 'explorer_maze2_N64_k4_L5'
with np.load(os.path.join(REPO, 'tests', 'golden', f + '.npz')) as z:
    r = {k: z[k] for k in z.files}
w = load_weights(ENVS[f.split('_')[1]]['ckpt'])
w64 = {k: (t.double() if t.is_floating_point() else t) for k, t in w.items()}
args64 = [torch.from_numpy(r[k]).double() if r[k].dtype.kind == 'f' else torch.from_numpy(r[k]) for k in ('v', 'goal', 'obstacles', 'edge_index')]
ref64 = torch.from_numpy(r['scores_fp64'])
ORIG_ATT = ref_cpu._attention; ORIG_MLP2 = ref_cpu._mlp2

def q32(x, on):          # "computed in fp32": inputs rounded, op in fp32, back to fp64
    return x.float() if on else x

def make_att(which, sw):
    def att(w_, pre, m, o, materialize=False):
        if not pre.startswith(which):
            return ORIG_ATT(w_, pre, m, o, materialize)
        d = m.shape[1]
        def lin(name, x, on):
            W = w_[pre + name + '.weight']
            return F.linear(x.float(), W.float()).double() if on else F.linear(x, W)
        mv = lin('.value', m, sw.get('v')); ov = lin('.value', o, sw.get('v'))
        mq = lin('.query', m, sw.get('qk')); mk = lin('.key', m, sw.get('qk')); ok = lin('.key', o, sw.get('qk'))
        if sw.get('logit'):
            obs = (mq.float() @ ok.float().T).double(); self_ = (mq.float() * mk.float()).sum(-1).double()
        else:
            obs = mq @ ok.T; self_ = (mq * mk).sum(-1)
        a = torch.cat((self_.unsqueeze(-1), obs), -1)
        if sw.get('softmax'):
            a = (a.float() / (d ** 0.5)).softmax(-1).double()
        else:
            a = (a / (d ** 0.5)).softmax(-1)
        if sw.get('pv'):
            new = (a[:, :1].float() * mv.float() + a[:, 1:].float() @ ov.float()).double()
        else:
            new = a[:, :1] * mv + a[:, 1:] @ ov
        x = new + m
        if sw.get('ln'):
            return ref_cpu._layer_norm(w, pre + '.layer_norm', x.float(), 1e-6).double()
        return ref_cpu._layer_norm(w_, pre + '.layer_norm', x, 1e-6)
    return att

def run(label):
    s = ref_cpu.explorer_forward(w64, *args64, int(r['loop']))
    e = (s - ref64).abs()
    print('%-50s rms %.3e max %.3e at %s' % (label, e.pow(2).mean().sqrt(), e.max(), torch.topk(e, 3).indices.tolist()))

for which in ('node_attentions.0', 'node_attentions'):
    for name, sw in (('q/k projections', dict(qk=1)), ('logits', dict(logit=1)), ('softmax', dict(softmax=1)), ('v proj', dict(v=1)),
                     ('PV', dict(pv=1)), ('LN', dict(ln=1)), ('qk+logit+softmax', dict(qk=1, logit=1, softmax=1)), ('v+pv+ln', dict(v=1, pv=1, ln=1))):
        ref_cpu._attention = make_att(which, sw)
        run('%s: %s in fp32' % (which, name))
ref_cpu._attention = ORIG_ATT
for enc in ('node_code', 'edge_code', 'node_free_code', 'edge_free_code', 'obs_node_code', 'obs_edge_code'):
    def mlp2(w_, name, x, enc=enc):
        if name == enc:
            return ORIG_MLP2(w, name, x.float()).double()
        return ORIG_MLP2(w_, name, x)
    ref_cpu._mlp2 = mlp2
    run('encoder %s in fp32' % enc)
    def mlp2b(w_, name, x, enc=enc):       # computed in fp64, result rounded to fp32
        y = ORIG_MLP2(w_, name, x)
        return y.float().double() if name == enc else y
    ref_cpu._mlp2 = mlp2b
    run('encoder %s fp64 compute, fp32 result' % enc)
ref_cpu._mlp2 = ORIG_MLP2
# logits magnitude
taps = {}
def att_spy(w_, pre, m, o, materialize=False):
    d = m.shape[1]
    mq = F.linear(m, w_[pre + '.query.weight']); ok = F.linear(o, w_[pre + '.key.weight'])
    a = (mq @ ok.T) / d ** 0.5
    print('   %-22s |m| max %.1f  |q| max %.1f  |k_obs| max %.1f  scaled logits: max|x| %.1f, row-range max %.1f' % (
        pre[:19], m.abs().max(), mq.abs().max(), ok.abs().max(), a.abs().max(), (a.max(1).values - a.min(1).values).max()))
    return ORIG_ATT(w_, pre, m, o, materialize)
ref_cpu._attention = att_spy
ref_cpu.explorer_forward(w64, *args64, int(r['loop']))

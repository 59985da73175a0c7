#!/usr/bin/env python
"""CPU experiment (no GPU): starting from an all-fp32 oracle run, which STAGES must be computed in fp64 to bring
max|scores - ref64| under 1e-5 on a full-size graph?  (the inverse of parity_sensitivity.py: the run is fp32, selected
function families are computed in fp64 and rounded to fp32 once).  Usage: parity_upgrade.py [env nodes k1 seed]"""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from oracle import ref_cpu
from gnnmp.synth import ENVS, synth_graph
from gnnmp.weights import load_weights

env, nodes, k1, seed = (sys.argv[1:5] + ['maze2', '1000', '8', str(1234 + 243)][len(sys.argv) - 1:])[:4]
g = synth_graph(env, int(nodes), int(k1), seed=int(seed))
w = load_weights(ENVS[env]['ckpt'])
w64 = {k: (t.double() if t.is_floating_point() else t) for k, t in w.items()}
a32 = [g['v'], g['goal'], g['obstacles'], g['edge_index']]
a64 = [t.double() if t.is_floating_point() else t for t in a32]
NAMES = dict(att='_attention', ff='_feed_forward', mlp2='_mlp2', lin='_lin', ln='_layer_norm')
ORIG = {k: getattr(ref_cpu, v) for k, v in NAMES.items()}
ref64 = ref_cpu.explorer_forward(w64, *a64, 5)


def in64(fn, match):
    def f(w_, name, *xs, **kw):
        if match(name):
            out = fn(w64, name, *[x.double() if torch.is_tensor(x) and x.is_floating_point() else x for x in xs], **kw)
            return tuple(o.float() for o in out) if isinstance(out, tuple) else out.float()
        return fn(w_, name, *xs, **kw)
    return f


def run(label, **patch):
    for k, v in ORIG.items():
        setattr(ref_cpu, NAMES[k], v)
    for k, m in patch.items():
        setattr(ref_cpu, NAMES[k], in64(ORIG[k], m))
    s = ref_cpu.explorer_forward(w, *a32, 5).double()
    e = (s - ref64).abs()
    print('%-78s rms %.3e max %.3e  #>1e-5 %d' % (label, e.pow(2).mean().sqrt(), e.max(), int((e > 1e-5).sum())))
    return e.max().item()


nf = lambda n: n == 'node_free_code'
nb0 = lambda n: n.startswith('node_attentions.0')
print('%s N=%s k1=%s seed %s: E = %d' % (env, nodes, k1, seed, g['edge_index'].shape[1]))
run('all fp32 (the reference\'s own fp32 run)')
run('round 3 GPU stretch: node_free_code + node block 0 attention in fp64', mlp2=nf, att=nb0)
run('+ node block 0 map_feed', mlp2=nf, att=nb0, ff=lambda n: n.startswith('node_attentions.0') and 'map_feed' in n)
run('+ node blocks 1, 2 attention', mlp2=nf, att=lambda n: n.startswith('node_att'))
run('+ whole node side (all node blocks: attention + map_feed, node_code)', mlp2=lambda n: n in ('node_free_code', 'node_code'), att=lambda n: n.startswith('node_att'), ff=lambda n: n.startswith('node_att'))
run('+ edge_free_code encoder', mlp2=lambda n: n in ('node_free_code', 'edge_free_code'), att=nb0)
run('+ edge block 0 attention', mlp2=nf, att=lambda n: nb0(n) or n.startswith('edge_attentions.0'))
run('+ edge_free_code + edge block 0 attention', mlp2=lambda n: n in ('node_free_code', 'edge_free_code'), att=lambda n: nb0(n) or n.startswith('edge_attentions.0'))
run('+ all edge attention (att only)', mlp2=nf, att=lambda n: nb0(n) or n.startswith('edge_att'))
run('+ loop lins (encoder / process / decoder)', mlp2=lambda n: nf(n) or n.startswith('process'), att=nb0, lin=lambda n: n.split('.')[0] in ('encoder', 'process', 'decoder'))
run('+ policy', mlp2=nf, att=nb0, lin=lambda n: n.startswith('policy'))
run('+ loop lins + policy', mlp2=lambda n: nf(n) or n.startswith('process'), att=nb0, lin=lambda n: n.split('.')[0] in ('encoder', 'process', 'decoder', 'policy'))
run('+ whole node side + loop lins + policy', mlp2=lambda n: n in ('node_free_code', 'node_code') or n.startswith('process'), att=lambda n: n.startswith('node_att'), ff=lambda n: n.startswith('node_att'), lin=lambda n: n.split('.')[0] in ('encoder', 'process', 'decoder', 'policy'))

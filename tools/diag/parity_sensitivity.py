#!/usr/bin/env python
"""Which STAGE of the forward, when computed in fp32 inside an otherwise fp64 run, produces the error of the sensitive
scores?  (oracle run in fp64 with one function family cast to fp32 and back)"""
import os, sys
import numpy as np, torch
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
ORIG = dict(att=ref_cpu._attention, ff=ref_cpu._feed_forward, mlp2=ref_cpu._mlp2, lin=ref_cpu._lin, ln=ref_cpu._layer_norm)

def in32(fn, match):
    def g(w_, name, *xs, **kw):
        if match(name):
            out = fn(w, name, *[x.float() if torch.is_tensor(x) and x.is_floating_point() else x for x in xs], **kw)
            return tuple(o.double() for o in out) if isinstance(out, tuple) else out.double()
        return fn(w_, name, *xs, **kw)
    return g

def run(label, **patch):
    for k, v in ORIG.items():
        setattr(ref_cpu, {'att': '_attention', 'ff': '_feed_forward', 'mlp2': '_mlp2', 'lin': '_lin', 'ln': '_layer_norm'}[k], v)
    for k, m in patch.items():
        name = {'att': '_attention', 'ff': '_feed_forward', 'mlp2': '_mlp2', 'lin': '_lin', 'ln': '_layer_norm'}[k]
        setattr(ref_cpu, name, in32(ORIG[k], m))
    s = ref_cpu.explorer_forward(w64, *args64, int(r['loop']))
    e = (s - ref64).abs()
    top = torch.topk(e, 3)
    print('%-44s rms %.3e max %.3e at %s' % (label, e.pow(2).mean().sqrt(), e.max(), top.indices.tolist()))

run('all fp64 (sanity)')
run('encoders (mlp2 *_code) in fp32', mlp2=lambda n: n.endswith('_code'))
run('node attention (att only) fp32', att=lambda n: n.startswith('node_att'))
run('edge attention (att only) fp32', att=lambda n: n.startswith('edge_att'))
for b in range(3):
    run('node att block %d fp32' % b, att=lambda n, b=b: n.startswith('node_attentions.%d' % b))
    run('edge att block %d fp32' % b, att=lambda n, b=b: n.startswith('edge_attentions.%d' % b))
run('node map_feed fp32', ff=lambda n: n.startswith('node_att') and 'map_feed' in n)
run('edge map_feed fp32', ff=lambda n: n.startswith('edge_att') and 'map_feed' in n)
run('obs_feed fp32', ff=lambda n: 'obs_feed' in n)
run('loop lins (encoder/process/decoder) fp32', lin=lambda n: n.split('.')[0] in ('encoder', 'process', 'decoder'), mlp2=lambda n: n.startswith('process'))
run('policy fp32', lin=lambda n: n.startswith('policy'))

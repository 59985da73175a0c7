import os, sys, subprocess
code = r"""
import os, sys, torch, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import gnnmp
from gnnmp.synth import ENVS, synth_batch_gpu
from gnnmp.weights import load_weights
e = ENVS['kuka7']
m = gnnmp.EncoderProcessDecoder(e['workspace'], e['C'], e['d'], e['S']).eval(); m.load_state_dict(load_weights(e['ckpt']))
graphs = synth_batch_gpu('kuka7', 700, 6, 40, 'cuda:0', seed0=77)
if os.environ.get('DBG_NOEDGES') == '1':
    for g in graphs: g['edge_index'] = g['edge_index'][:, :0]
if os.environ.get('DBG_NOEDGES') == '2':
    for g in graphs: g['edge_index'] = g['edge_index'][:, g['edge_index'][1] < 350]
b = gnnmp.GraphBatch.from_graphs(graphs, e['S'], 'cuda:0')
for loop in (1, 2):
    s = m.forward_batch(b, loop)
    h = m.debug_tap(b, 1); dec = m.debug_tap(b, 2)
    torch.cuda.synchronize()
    np.save('/tmp/w8dbg_%s_l%d_s.npy' % (os.environ['GNNMP_MP_W8'], loop), s.cpu().numpy())
    np.save('/tmp/w8dbg_%s_l%d_h.npy' % (os.environ['GNNMP_MP_W8'], loop), h.cpu().numpy())
    np.save('/tmp/w8dbg_%s_l%d_d.npy' % (os.environ['GNNMP_MP_W8'], loop), dec.cpu().numpy())
"""
for w8 in ('0', '1'):
    r = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, GNNMP_MP_W8=w8), capture_output=True, text=True)
    if r.returncode: print(r.stderr[-1500:])
import numpy as np
for loop in (1, 2):
    for what in ('h', 'd', 's'):
        a = np.load('/tmp/w8dbg_0_l%d_%s.npy' % (loop, what)); b = np.load('/tmp/w8dbg_1_l%d_%s.npy' % (loop, what))
        bad = a != b
        print('loop', loop, what, a.shape, 'differing elements', int(bad.sum()), 'of', a.size, 'max|d| %.3g' % np.abs(a - b).max())
        if a.ndim == 2 and bad.any():
            rows = np.nonzero(bad.any(1))[0]; cols = np.nonzero(bad.any(0))[0]
            print('   rows differing', len(rows), 'first', rows[:12], ' rows mod 32:', np.bincount(rows % 32, minlength=32).tolist())
            print('   cols differing', len(cols), cols[:70].tolist())

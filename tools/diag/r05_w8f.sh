#!/bin/bash
mkdir -p gpurun_out/r05
{
python tools/diag/abx.py 3f base,GNNMP_MP_W8=0 base
python tools/diag/abx.py 3 base,GNNMP_MP_W8=0 base
python - <<'PY'
import os, sys, subprocess
code = r"""
import os, sys, torch, hashlib
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import gnnmp
from gnnmp.synth import ENVS, synth_batch_gpu
from gnnmp.weights import load_weights
e = ENVS['kuka7']
m = gnnmp.EncoderProcessDecoder(e['workspace'], e['C'], e['d'], e['S']).eval(); m.load_state_dict(load_weights(e['ckpt']))
for dt in ('fp32', 'bf16'):
    m.mlp_dtype = dt
    for (n, k, g) in ((2000, 10, 64), (700, 6, 40), (300, 5, 200), (2000, 10, 3)):
        graphs = synth_batch_gpu('kuka7', n, k, g, 'cuda:0', seed0=77)
        b = gnnmp.GraphBatch.from_graphs(graphs, e['S'], 'cuda:0')
        for loop in (1, 5):
            s = m.forward_batch(b, loop)
            torch.cuda.synchronize()
            print('%s %d %d %d loop %d: %s sum %.6f finite %s' % (dt, n, k, g, loop, hashlib.sha256(s.cpu().numpy().tobytes()).hexdigest()[:16], float(s.double().sum()), bool(torch.isfinite(s).all())))
"""
res = {}
for w8 in ('0', '1'):
    r = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, GNNMP_MP_W8=w8), capture_output=True, text=True)
    res[w8] = r.stdout
    print('GNNMP_MP_W8=%s rc=%d\n%s%s' % (w8, r.returncode, r.stdout, r.stderr[-800:] if r.returncode else ''))
print('BIT-IDENTICAL' if res['0'] == res['1'] and res['0'] else 'DIFFERENT')
PY
} > gpurun_out/r05/w8f.txt 2>&1
cat gpurun_out/r05/w8f.txt

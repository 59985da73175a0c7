"""single-graph explorer forward: kernel durations vs wall (run under rocprofv3 --kernel-trace --stats)"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import torch, gnnmp
from gnnmp.weights import load_weights
from gnnmp.synth import ENVS, synth_graph
dev = torch.device('cuda:0')
env = sys.argv[1] if len(sys.argv) > 1 else 'maze2'
n, k = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1000, 8)
e = ENVS[env]
m = gnnmp.EncoderProcessDecoder(e['workspace'], e['C'], e['d'], e['S']).eval()
m.load_state_dict(load_weights(e['ckpt']))
if len(sys.argv) > 4 and sys.argv[4] != '-':
    m.mlp_dtype = sys.argv[4]
if len(sys.argv) > 5 and sys.argv[5] == 'noobs':
    m.use_obstacles = False
g = {kk: (v.to(dev) if torch.is_tensor(v) else v) for kk, v in synth_graph(env, n, k).items()}
b = m._single(g['goal'], g['v'], g['obstacles'], g['edge_index'])
for _ in range(20):
    m.forward_batch(b, 5)
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 200
for _ in range(N):
    m.forward_batch(b, 5)
torch.cuda.synchronize()
print('prebuilt batch forward: %.1f us per call' % ((time.perf_counter() - t0) / N * 1e6))

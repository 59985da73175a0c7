import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import gnnmp
from gnnmp.synth import synth_graph
from gnnmp.weights import load_weights
DEV = 'cuda:0'
graphs = [{k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in synth_graph('maze2', 150 + 30 * i, 5, seed=20 + i).items()} for i in range(4)]
m = gnnmp.EncoderProcessDecoder(2, 2, 32, 2); m.load_state_dict(load_weights('weights_maze')); m.train()
batch = gnnmp.GraphBatch.from_graphs(graphs, 2, DEV)
coef = torch.linspace(-1, 1, batch.total_edges, device=DEV)
runs = []
for _ in range(3):
    m.zero_grad()
    s = m.train_scores(batch, 3)
    (s * coef).sum().backward()
    runs.append((s.detach().clone(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}))
print('scores equal', torch.equal(runs[0][0], runs[1][0]), torch.equal(runs[1][0], runs[2][0]))
for n in runs[0][1]:
    a, b, c = runs[0][1][n], runs[1][1][n], runs[2][1][n]
    print('%-32s max|g| %.3e  |r0-r1| %.3e  |r1-r2| %.3e' % (n, a.abs().max(), (a - b).abs().max(), (b - c).abs().max()))

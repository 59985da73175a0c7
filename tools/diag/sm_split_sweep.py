"""smoother batch rate with the split (tile per workgroup) vs wave (tile per wave) kernels: GNNMP_SM_SPLIT=0/1 python ..."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import torch, gnnmp
from gnnmp.weights import load_weights
from gnnmp.planner import chain_edge_index
from gnnmp.smoother import SmoothBatch
dev = torch.device('cuda:0')
ms = gnnmp.ModelSmoother(workspace_size=2, config_size=2, embed_size=128, obs_size=6).eval()
ms.load_state_dict(load_weights('smooth_2d_attv3'))
if len(sys.argv) > 1:
    ms.mlp_dtype = sys.argv[1]
gen = torch.Generator().manual_seed(3)
P, F, Co = 20, 500, 500
for B in ([int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else (1, 4, 16, 32, 56, 64, 128, 256)):
    ps = [(torch.rand(P, 2, generator=gen) * 2 - 1) for _ in range(B)]
    fs = [(torch.rand(F, 2, generator=gen) * 2 - 1) for _ in range(B)]
    cs = [(torch.rand(Co, 2, generator=gen) * 2 - 1) for _ in range(B)]
    sb = SmoothBatch(ps, fs, cs, [chain_edge_index(P)] * B, dev)
    for _ in range(5):
        ms.forward_batch(sb, 1)
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            ms.forward_batch(sb, 1)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 20)
    print('B=%-4d %.1f us' % (B, sorted(ts)[2] * 1e6))

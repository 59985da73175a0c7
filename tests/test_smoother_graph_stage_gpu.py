"""The smoother's one-launch graph stage (kNN + coalesced edge list in one workgroup per problem, sm_graph_kernel) against the
two-launch form it replaced (sm_knn_kernel + sm_edges_kernel, GNNMP_SM_FUSED_GRAPH=0): same neighbour sets and edge lists, hence
the same bits in the smoothed paths.  The switch is read once per process, so the forced run happens in a subprocess.
Shapes: ragged batches (2 to 45 waypoints, 0 to 1100 samples per problem, fewer samples than k, duplicate and out-of-range
caller edges, 2 / 7 / 14 coordinates), a problem given by its totals alone, 2048 samples per problem (`wide`: the 32-slot
instantiation), and a batch whose buffers exceed the LDS share of a workgroup (`big`: both runs take the two-launch form there --
it must still be reachable and agree)."""
import os
import subprocess
import sys

import pytest
import torch

from conftest import REPO

pytestmark = pytest.mark.gpu

SCRIPT = r'''
import sys, torch
sys.path.insert(0, %r); sys.path.insert(0, %r + '/tests')
import gnnmp
from gnnmp.weights import load_weights
from gnnmp.planner import chain_edge_index
case, out = sys.argv[1], sys.argv[2]
C = {'c2': 2, 'c7': 7, 'c14': 14, 'wide': 7, 'big': 14, 'single': 7}[case]
name = {2: 'smooth_2d_attv3', 7: 'smooth_7d_attv3', 14: 'smooth_14d_attv3'}[C]
m = gnnmp.ModelSmoother(workspace_size=3, config_size=C, embed_size=128, obs_size=6).eval()
m.load_state_dict(load_weights(name))
gen = torch.Generator().manual_seed(101 + C)
mk = lambda n: torch.rand(n, C, generator=gen) * 2 - 1
def edges(P, i):
    e = chain_edge_index(P)
    if i %% 3 == 1:      # duplicates, a self loop, sources that are sample nodes, one out-of-range pair (dropped like the kNN's -1)
        extra = torch.tensor([[0, 1, 1, P + 3, P - 1, 10 ** 6], [1, 0, 1, 0, P - 1, 0]], dtype=torch.long)
        e = torch.cat([e, extra], dim=1)
    return e
res = {}
if case == 'single':
    path, free, coll = mk(20), mk(500), mk(500)
    res['one'] = m(path=path.cuda(), free=free.cuda(), collided=coll.cuda(), obstacles=None,
                   edge_index=chain_edge_index(20).cuda(), loop=3).cpu()
else:
    if case in ('wide', 'big'):
        sizes = [(400, 1100, 948), (380, 900, 1100)]
    else:
        sizes = [(2, 0, 0), (3, 4, 3), (20, 500, 500), (45, 700, 400), (7, 9, 0), (31, 0, 64), (33, 1000, 24), (12, 5, 4)]
    paths = [mk(s[0]) for s in sizes]
    frees = [mk(s[1]) for s in sizes]
    colls = [mk(s[2]) for s in sizes]
    sb = gnnmp.SmoothBatch(paths, frees, colls, [edges(s[0], i) for i, s in enumerate(sizes)], 'cuda:0')
    for loop in (1, 2):
        res['loop%%d' %% loop] = m.forward_batch(sb, loop).cpu()
torch.save(res, out)
'''


@pytest.mark.parametrize('case', ['c2', 'c7', 'c14', 'single', 'wide', 'big'])
def test_one_launch_graph_stage_equals_two_launch_form_bitwise(case, tmp_path):
    outs = []
    for fused in ('1', '0'):
        out = str(tmp_path / ('out_%s.pt' % fused))
        env = dict(os.environ, GNNMP_SM_FUSED_GRAPH=fused)
        subprocess.run([sys.executable, '-c', SCRIPT % (REPO, REPO), case, out], check=True, env=env, timeout=300)
        outs.append(torch.load(out))
    assert outs[0].keys() == outs[1].keys()
    for k in outs[0]:
        assert bool(torch.isfinite(outs[0][k]).all())
        assert torch.equal(outs[0][k], outs[1][k]), k


@pytest.mark.parametrize('case', ['c7', 'c14'])
def test_message_kernel_without_target_role_same_bits(case, tmp_path):
    """The split message kernel's edge tiles normally continue from the target rows its first workgroups publish; the path
    that computes the target half inside the edge tile (taken when a flag does not show up within the polling budget, and
    with GNNMP_SM_NO_TARGET_ROLE=1) must give the same bits."""
    outs = []
    for no_target in ('0', '1'):
        out = str(tmp_path / ('out_%s.pt' % no_target))
        env = dict(os.environ, GNNMP_SM_NO_TARGET_ROLE=no_target)
        subprocess.run([sys.executable, '-c', SCRIPT % (REPO, REPO), case, out], check=True, env=env, timeout=300)
        outs.append(torch.load(out))
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), k

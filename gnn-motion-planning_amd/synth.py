"""Synthetic random-geometric-graph inputs for tests and ``bench.py``.

There are no datasets on the GPU box, so the benchmark workload is generated: per graph g
(seed ``1234 + g``, CPU ``torch.Generator``, float64 draws rounded to fp32)

  * ``v = (U[0,1)^{N x C} * 2 - 1) * lim`` with ``lim`` the environment's joint box;
    ``v`` is the FIRST draw from the generator; rows ``0 .. N/2-1`` play "free" samples
    (row 0 = start, row 1 = goal), the rest "collided"; ``goal = v[1]``.
  * ``edge_index`` = the reference's graph rule (eval_gnn.py:160-164) with k1 given directly.
  * obstacles: maze / snake -> ``O = 116`` distinct cells of a 15 x 15 grid as
    ``(i, j) / 15 - 0.5`` (maze_env.py:73-79); kuka / ur5 -> ``O = 5`` boxes
    ``(halfExtents ~ U(0.05, 0.3)^3, basePosition ~ U(-1, 1)^3)`` flattened to 6 numbers.
"""
import math

import torch

from .graph_build import build_edges

_KUKA7 = [2.96706, 2.09440, 2.96706, 2.09440, 2.96706, 2.09440, 3.05433]

ENVS = {
    # name: (config_size C, embed d, obs_size S, joint box, obstacle kind, checkpoint)
    'maze2': dict(C=2, d=32, S=2, lim=[1.0, 1.0], obs='grid', ckpt='weights_maze', workspace=2),
    'maze3': dict(C=3, d=32, S=2, lim=[1.0, 1.0, 0.4], obs='grid', ckpt='weights_maze_3', workspace=2),
    'kuka7': dict(C=7, d=64, S=6, lim=_KUKA7, obs='box', ckpt='weights_kuka', workspace=3),
    'ur5': dict(C=6, d=32, S=6, lim=[2 * math.pi, 2 * math.pi, math.pi] + [2 * math.pi] * 3,
                obs='box', ckpt='weights_ur5', workspace=3),
    'snake7': dict(C=7, d=32, S=2, lim=[9.0, 9.0] + [math.pi] * 5, obs='grid', ckpt='weights_snake',
                   workspace=3),
    'kuka13': dict(C=13, d=32, S=6, lim=_KUKA7 + _KUKA7[:6], obs='box', ckpt='weights_kuka_13',
                   workspace=3),
    'kuka14': dict(C=14, d=32, S=6, lim=_KUKA7 + _KUKA7, obs='box', ckpt='kuka_14', workspace=3),
}


def synth_graph(env, n_nodes, k1, seed=1234, n_obs=None):
    """One synthetic planning graph; returns dict(v, goal, obstacles, edge_index, n_free)."""
    e = ENVS[env]
    gen = torch.Generator().manual_seed(int(seed))
    lim = torch.tensor(e['lim'], dtype=torch.float64)
    v = ((torch.rand(n_nodes, e['C'], generator=gen, dtype=torch.float64) * 2 - 1) * lim).float()
    n_free = n_nodes // 2
    if e['obs'] == 'grid':
        O = 116 if n_obs is None else n_obs
        cells = torch.randperm(225, generator=gen)[:O].sort().values
        obstacles = (torch.stack((cells // 15, cells % 15), dim=1).to(torch.float64) / 15 - 0.5).float()
    else:
        O = 5 if n_obs is None else n_obs
        half = torch.rand(O, 3, generator=gen, dtype=torch.float64) * 0.25 + 0.05
        base = torch.rand(O, 3, generator=gen, dtype=torch.float64) * 2 - 1
        obstacles = torch.stack((half, base), dim=1).float()          # [O, 2, 3] like kuka_env.py:98
    return {'v': v, 'goal': v[1].clone(), 'obstacles': obstacles,
            'edge_index': build_edges(v, n_free, k1), 'n_free': n_free}


def synth_batch_gpu(env, n_nodes, k1, n_graphs, device, seed0=1234, n_obs=None):
    """``n_graphs`` synthetic graphs (seeds ``seed0 + i``) with the SAME node / obstacle draws as
    :func:`synth_graph`, but with the kNN edge construction done in one batched call on the GPU
    (graph_kernels.hip; bit-identical edge_index to the host builder, tests/test_graph_build_gpu.py).
    Returns a list of graph dicts whose tensors live on ``device``."""
    from .graph_build import build_edges_gpu
    e = ENVS[env]
    vs, obs = [], []
    for i in range(n_graphs):
        gen = torch.Generator().manual_seed(int(seed0 + i))
        lim = torch.tensor(e['lim'], dtype=torch.float64)
        vs.append(((torch.rand(n_nodes, e['C'], generator=gen, dtype=torch.float64) * 2 - 1) * lim).float())
        if e['obs'] == 'grid':
            O = 116 if n_obs is None else n_obs
            cells = torch.randperm(225, generator=gen)[:O].sort().values
            obs.append((torch.stack((cells // 15, cells % 15), dim=1).to(torch.float64) / 15 - 0.5).float())
        else:
            O = 5 if n_obs is None else n_obs
            half = torch.rand(O, 3, generator=gen, dtype=torch.float64) * 0.25 + 0.05
            base = torch.rand(O, 3, generator=gen, dtype=torch.float64) * 2 - 1
            obs.append(torch.stack((half, base), dim=1).float())
    v_all = torch.cat(vs).to(device)
    ptr = (torch.arange(n_graphs + 1, dtype=torch.int64) * n_nodes).to(torch.int32).to(device)
    ei, eptr = build_edges_gpu(v_all, ptr, [n_nodes // 2] * n_graphs, [k1] * n_graphs)
    eptr = eptr.cpu().tolist()
    out = []
    for i in range(n_graphs):
        vd = v_all[i * n_nodes:(i + 1) * n_nodes]
        out.append({'v': vd, 'goal': vd[1].clone(), 'obstacles': obs[i].to(device),
                    'edge_index': ei[:, eptr[i]:eptr[i + 1]], 'n_free': n_nodes // 2})
    return out

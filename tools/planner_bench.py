#!/usr/bin/env python
"""Whole-planner throughput on 2-D maze problems (informational; the headline metric is bench.py).

The GNN forwards run on the GPU; sampling, graph construction (host or device), the greedy frontier loop
and every collision check run on ONE host core in this process -- exactly the split north_star
describes.  Problems come from tests/golden/evalset_*.npz (the first problems of the reference's
mazes_hard.npz), cycled.  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import gnnmp  # noqa: E402
from conftest import golden_files  # noqa: E402
from gnnmp.weights import load_weights  # noqa: E402
from gnnmp import planner  # noqa: E402
from gnnmp.maze2d import Maze2D  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--problems', type=int, default=48)
    ap.add_argument('--sparse', action='store_true', help='sparse frontier on per-edge scores instead of the dense N x N matrix')
    ap.add_argument('--gpu-graph', action='store_true', help='build the kNN graph on the device')
    ap.add_argument('--device-explore', action='store_true',
                    help='explore stage of ALL problems in one device pass (graphs, forward, greedy loop, collision checks)')
    ap.add_argument('--device-smooth', action='store_true',
                    help='with --device-explore: the smoothing stage too (batched smoother forwards + steering on the device)')
    ap.add_argument('--device-eval', action='store_true',
                    help='planner.eval_gnn_device at its defaults (passes of 128 problems on 2 worker threads / streams, sampling '
                         'ahead on its own thread): the evaluation loop a user runs; median of 3 passes over --problems')
    ap.add_argument('--batch', type=int, default=500)
    ap.add_argument('--k', type=int, default=30)
    a = ap.parse_args()
    with np.load(golden_files('evalset_mazehard_first12')[0]) as f:
        maps, init, goal = f['maps'], f['init_states'], f['goal_states']
    env = Maze2D(maps, init, goal)
    dev = 'cuda:0'
    m = gnnmp.EncoderProcessDecoder(2, 2, 32, 2).eval()
    m.load_state_dict(load_weights('weights_maze'))
    ms = gnnmp.ModelSmoother(workspace_size=2, config_size=2, embed_size=128, obs_size=6).eval()
    ms.load_state_dict(load_weights('smooth_2d_attv3'))
    if a.device_eval:
        idx = [i % maps.shape[0] for i in range(a.problems)]
        planner.eval_gnn_device(env, idx, m, ms, batch=a.batch, k=a.k, device=dev)               # warm-up (allocator pools of the worker streams)
        walls = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = planner.eval_gnn_device(env, idx, m, ms, batch=a.batch, k=a.k, device=dev)
            torch.cuda.synchronize()
            walls.append(time.perf_counter() - t0)
        wall = sorted(walls)[1]
        print(json.dumps({'problems': a.problems, 'success': int(out[0]), 'stage': 'planner.eval_gnn_device (whole planner, defaults: passes of 128, 2 workers)',
                          'problems_per_s': round(a.problems / wall, 1), 'three_passes': [round(a.problems / w, 1) for w in walls],
                          'collision_checks_total': round(out[1], 2), 'collision_checks_explore': round(out[7], 2),
                          'path_cost': round(out[3], 4), 'host_threads': 'main + 1 sampler + 2 device-pass workers',
                          'config': 'maze2 hard, batch=%d, k=%d, smoothing on' % (a.batch, a.k)}))
        return
    if a.device_explore:
        probs = [dict(map=maps[i % maps.shape[0]], init_state=init[i % maps.shape[0]], goal_state=goal[i % maps.shape[0]])
                 for i in range(a.problems)]
        np.random.seed(1234)
        sm = ms if a.device_smooth else None
        np.random.seed(1234)
        planner.explore_maze_batch(probs, m, dev, batch=a.batch, k=a.k, model_s=sm)              # warm-up at full size (allocator)
        np.random.seed(1234)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = planner.explore_maze_batch(probs, m, dev, batch=a.batch, k=a.k, model_s=sm)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        tms = {}
        np.random.seed(1234)
        planner.explore_maze_batch(probs, m, dev, batch=a.batch, k=a.k, model_s=sm, timings=tms)     # second pass with syncs
        out = {'problems': a.problems, 'success': sum(int(x['success']) for x in res),
               'stage': 'explore + smoothing (whole planner)' if sm else 'explore only (no smoothing)',
               'problems_per_s': round(a.problems / wall, 2), 's_per_problem': round(wall / a.problems, 5),
               'collision_checks_explore': round(sum(x['c_explore'] for x in res) / a.problems, 2),
               'host_cores_used': 1, 'host_work': 'rejection sampling only (vectorised, same numpy stream)',
               'device_work': 'kNN graphs, explorer forward, greedy frontier, collision checks' +
                              (', 5 x (smoother forward, steering)' if sm else ''),
               'config': 'maze2 hard, batch=%d, k=%d' % (a.batch, a.k),
               'stage_ms_per_problem': {k_: round(1e3 * v_ / a.problems, 4) for k_, v_ in tms.items()}}
        if sm:
            out['collision_checks_total'] = round(sum(x['c_explore'] + x['c_smooth'] for x in res) / a.problems, 2)
            out['path_cost'] = round(float(np.mean([planner.path_cost(x['smooth_path']) for x in res if x['success']])), 4)
        print(json.dumps(out))
        return
    np.random.seed(1234)
    torch.manual_seed(1234)
    env.init_new_problem(0)
    planner.explore(env, m, ms, True, batch=a.batch, t_max=500, k=a.k, device=dev, sparse=a.sparse, gpu_graph=a.gpu_graph)     # warm-up
    tot = dict(success=0, forward=0.0, total=0.0, explore=0.0, c_explore=0, c_smooth=0)
    t0 = time.perf_counter()
    for i in range(a.problems):
        env.init_new_problem(i % maps.shape[0])
        r = planner.explore(env, m, ms, True, batch=a.batch, t_max=500, k=a.k, device=dev, sparse=a.sparse,
                            gpu_graph=a.gpu_graph)
        tot['success'] += int(r['success']); tot['forward'] += r['forward']; tot['total'] += r['total']
        tot['explore'] += r['total_explore']; tot['c_explore'] += r['c_explore']; tot['c_smooth'] += r['c_smooth']
    wall = time.perf_counter() - t0
    n = a.problems
    print(json.dumps({'problems': n, 'success': tot['success'], 'problems_per_s': round(n / wall, 3),
                      's_per_problem': round(wall / n, 4), 'gnn_forward_s_per_problem': round(tot['forward'] / n, 5),
                      'host_s_per_problem': round((tot['total'] - tot['forward']) / n, 4),
                      'collision_checks_explore': round(tot['c_explore'] / n, 2),
                      'collision_checks_total': round((tot['c_explore'] + tot['c_smooth']) / n, 2),
                      'host_cores_used': 1, 'graph_build': 'device' if a.gpu_graph else 'host', 'frontier': 'sparse heap' if a.sparse else 'dense N x N (reference form)',
                      'config': 'maze2 hard, batch=%d, k=%d, smoothing on' % (a.batch, a.k)}))


if __name__ == '__main__':
    main()

#!/usr/bin/env python
"""Evaluate a 2-D maze problem set with the device planner -- the counterpart of the reference's
``eval_gnn(...)`` driver call (eval_gnn.py:96-145; notebook cell main.ipynb:79) -- on one GPU or, under
``python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/eval_maze.py ...``, sharded over N
GPUs: every rank plans a contiguous block of the problems (starting at the RNG position the sequential run would
have reached), the per-problem rows are gathered with RCCL, rank 0 prints the reference's aggregate lines.

  python tools/eval_maze.py --problems tests/golden/evalset_mazehard_first1000.npz
  (any .npz with maps [n, w, w], init_states [n, 2], goal_states [n, 2], e.g. the reference's maze_files/mazes_hard.npz)"""
import argparse
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import gnnmp  # noqa: E402
from gnnmp.weights import load_weights  # noqa: E402
from gnnmp import planner  # noqa: E402
from gnnmp.dist import gather_problem_results  # noqa: E402
from gnnmp.maze2d import Maze2D  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--problems', default=os.path.join(REPO, 'tests', 'golden', 'evalset_mazehard_first1000.npz'))
    ap.add_argument('--count', type=int, default=0, help='first COUNT problems (0 = all)')
    ap.add_argument('--seed', type=int, default=1234)
    ap.add_argument('--batch', type=int, default=500)
    ap.add_argument('--k', type=int, default=30)
    ap.add_argument('--chunk', type=int, default=128, help='problems per device pass')
    ap.add_argument('--workers', type=int, default=2, help='host threads driving device passes')
    a = ap.parse_args()
    rank, local, world = (int(os.environ.get(k, d)) for k, d in (('RANK', 0), ('LOCAL_RANK', 0), ('WORLD_SIZE', 1)))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:
        dist.init_process_group('nccl', device_id=dev)
    with np.load(a.problems) as f:
        env = Maze2D(f['maps'], f['init_states'], f['goal_states'])
    n = env.size if a.count <= 0 else min(a.count, env.size)
    m = gnnmp.EncoderProcessDecoder(2, 2, 32, 2).eval()
    m.load_state_dict(load_weights('weights_maze'))
    ms = gnnmp.ModelSmoother(workspace_size=2, config_size=2, embed_size=128, obs_size=6).eval()
    ms.load_state_dict(load_weights('smooth_2d_attv3'))
    rows = []
    t0 = time.perf_counter()
    planner.eval_gnn_device(env, range(n), m, ms, seed=a.seed, batch=a.batch, k=a.k, device=dev, chunk=a.chunk,
                            workers=a.workers, rows_out=rows, shard=(rank, world) if world > 1 else None)
    local_rows = torch.tensor(np.array(rows, dtype=np.float64).reshape(-1, 7), device=dev)
    allrows = gather_problem_results(local_rows).cpu().numpy()
    if world > 1:
        dist.barrier()
    wall = time.perf_counter() - t0
    if rank == 0:
        ok = allrows[:, 0] > 0
        print('problems %d on %d GPU(s): %.2f s wall (%.0f problems/s)' % (allrows.shape[0], world, wall, allrows.shape[0] / wall))
        print('success rate: %d / %d' % (int(ok.sum()), allrows.shape[0]))                       # eval_gnn.py:131-139
        print('collision check: %.2f' % (allrows[:, 3] + allrows[:, 4]).mean())
        print('collision check (explore): %.2f' % allrows[:, 3].mean())
        print('path cost: %.4f -> smoothed %.4f' % (allrows[ok, 1].mean(), allrows[ok, 2].mean()))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

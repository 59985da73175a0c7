#!/usr/bin/env python
"""Run ON THE GPU BOX: BASELINE configs[4] as ONE job on one GPU -- 32 kuka14 problems with 5000-node k1 = 16 RGGs:
explorer forward (bf16 operands, kuka_14 checkpoint) -> a waypoint path per problem read off the explorer's scores -> the
smoothing stage's 5 x ModelSmoother.forward(loop = 1) (bf16 operands, smooth_14d_attv3 checkpoint, d = 128), everything
enqueued on one stream with no host synchronisation in between.

Reference shape of the stage chain: eval_gnn.py:249-258 (explore -> model_smooth) and smoother.py:233-246 (five iterations of
smoother forward + steering).  The 14-DoF environment's collision checks are PyBullet calls that stay on the host by
north_star, and PyBullet is not in this image, so the job is synthetic in exactly two places, both stated here:
  * the path: waypoint 0 is the goal node (row 1); waypoint k + 1 is the SOURCE of the best-scored incoming edge of waypoint k
    (lowest column on ties) -- a deterministic function of the explorer's scores computed with torch ops on the device, so
    the smoother really consumes the explorer's output; problem b's path has P_b = 5 + (7 b mod 31) waypoints (5 .. 35);
  * the steering: every proposal is accepted (new path = the smoother's output), i.e. no collision-checked steering
    between the five smoother calls.
Samples handed to the smoother: the first 500 free and the first 500 collided rows of the problem (smoother.py:57-58).

    python tools/cfg5_pipeline.py [--problems 32] [--reps 20]     -> one JSON line (problems/s, stage split)
"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import gnnmp  # noqa: E402
from gnnmp.smoother import SmoothBatch  # noqa: E402
from gnnmp.synth import ENVS, synth_batch_gpu  # noqa: E402
from gnnmp.weights import load_weights  # noqa: E402

P_MAX = 35


class Cfg5Job:
    def __init__(self, n_problems=32, nodes=5000, k1=16, device='cuda:0', seed0=1234, mlp_dtype='bf16', env='kuka14',
                 smoother='smooth_14d_attv3'):
        e = ENVS[env]
        self.dev, self.B, self.loop = torch.device(device), n_problems, 5
        self.graphs = synth_batch_gpu(env, nodes, k1, n_problems, device, seed0=seed0)
        self.model = gnnmp.EncoderProcessDecoder(e['workspace'], e['C'], e['d'], e['S']).eval()
        self.model.load_state_dict(load_weights(e['ckpt']), strict=True)
        self.model.mlp_dtype = mlp_dtype
        self.smoother = gnnmp.ModelSmoother(workspace_size=e['workspace'], config_size=e['C'], embed_size=128, obs_size=6).eval()
        self.smoother.load_state_dict(load_weights(smoother), strict=True)
        self.smoother.mlp_dtype = mlp_dtype
        self.batch = gnnmp.GraphBatch.from_graphs(self.graphs, e['S'], device)
        nptr = self.batch.node_ptr.to(torch.int64)
        ecnt = (self.batch.edge_ptr[1:] - self.batch.edge_ptr[:-1]).to(torch.int64)
        eoff = torch.repeat_interleave(nptr[:-1], ecnt)
        self.src = self.batch.edge_index[0] + eoff                 # global node ids
        self.tgt = self.batch.edge_index[1] + eoff
        self.col = torch.arange(self.batch.total_edges, device=device)
        self.goal_nodes = nptr[:-1] + 1                            # row 1 of every graph (gnnmp.synth)
        self.counts = [5 + (7 * b) % 31 for b in range(n_problems)]
        take = torch.zeros(n_problems, P_MAX, dtype=torch.bool)
        for b, c in enumerate(self.counts):
            take[b, :c] = True
        self.take_idx = take.reshape(-1).nonzero().reshape(-1).to(device)      # host-side: no boolean-mask sync in the job
        n_free = [g['n_free'] for g in self.graphs]
        nfree_s = [min(500, nf) for nf in n_free]
        ncoll_s = [min(500, int(g['v'].shape[0]) - nf) for g, nf in zip(self.graphs, n_free)]
        self.free = torch.cat([g['v'][:k] for g, k in zip(self.graphs, nfree_s)]).contiguous()
        self.coll = torch.cat([g['v'][nf:nf + k] for g, nf, k in zip(self.graphs, n_free, ncoll_s)]).contiguous()
        from gnnmp.planner import chain_edge_index
        eis = [chain_edge_index(c) for c in self.counts]
        self.chain = torch.cat(eis, dim=1).to(device)
        self.chain_counts = [int(x.shape[1]) for x in eis]
        self.free_counts, self.coll_counts = nfree_s, ncoll_s
        # the smoothing batch descriptor (prefix arrays on the device) is the same for all five iterations: built once, only
        # its path rows change (building it per iteration cost four small blocking host-to-device copies each time)
        self.sb = SmoothBatch.from_device(torch.zeros(sum(self.counts), e['C'], device=device), self.free, self.coll, self.chain,
                                          self.counts, self.free_counts, self.coll_counts, self.chain_counts)

    # ---- the three stages; each only enqueues work on the current stream
    def explore(self):
        return self.model.forward_batch(self.batch, self.loop)

    def paths(self, scores):
        """[sum P_b, C] waypoint rows: the chain of best-scored incoming edges from the goal node (see the module docstring).
        All torch ops on the current stream, no host synchronisation: one 64-bit scatter-max picks every node's best incoming
        edge (key = order-preserving integer image of the score, then the LOWEST column), and the chain is followed with jump
        tables (J_2s = J_s o J_s): 11 small launches instead of one gather per waypoint."""
        n, ncol = self.batch.total_nodes, self.col.numel()
        bits = scores.view(torch.int32)
        ordered = (bits ^ ((bits >> 31) & 0x7fffffff)).to(torch.int64)      # monotone in the float value
        key = (ordered << 32) | (0xffffffff - self.col)                    # ties: the lowest column wins
        best = torch.full((n,), torch.iinfo(torch.int64).min, dtype=torch.int64, device=self.dev)
        best.scatter_reduce_(0, self.tgt, key, 'amax', include_self=True)
        first = (0xffffffff - (best & 0xffffffff)).clamp_(max=ncol - 1)   # every node has its self loop, so this is a real column
        jump = self.src[first]                                             # J_1: node -> source of its best incoming edge
        pos = torch.empty(self.B, 64, dtype=torch.int64, device=self.dev)
        pos[:, 0] = self.goal_nodes
        s = 1
        while s < P_MAX:
            pos[:, s:2 * s] = jump[pos[:, 0:s]]                            # waypoints s .. 2 s - 1 are s steps behind waypoints 0 .. s - 1
            s *= 2
            if s < P_MAX:
                jump = jump[jump]
        nodes = pos[:, :P_MAX].reshape(-1)[self.take_idx]                  # problem after problem, waypoint after waypoint
        return self.batch.v[nodes].contiguous()

    def smooth(self, path, iters=5):
        for _ in range(iters):                                              # smoother.py:233-246 with every proposal accepted
            self.sb.path = path
            path = self.smoother.forward_batch(self.sb, 1)
        return path

    def run(self):
        """The pipelined job: nothing but enqueues between the first kernel of the explorer and the last of the smoother."""
        return self.smooth(self.paths(self.explore()))

    def run_separate(self):
        """The same stages as separate, synchronised calls on cloned intermediates (what the tests compare run() with)."""
        s = self.explore().clone()
        torch.cuda.synchronize(self.dev)
        p = self.paths(s).clone()
        torch.cuda.synchronize(self.dev)
        for _ in range(5):
            p = self.smooth(p, iters=1).clone()
            torch.cuda.synchronize(self.dev)
        return p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--problems', type=int, default=32)
    ap.add_argument('--nodes', type=int, default=5000)
    ap.add_argument('--k1', type=int, default=16)
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--mlp-dtype', default='bf16')
    a = ap.parse_args()
    job = Cfg5Job(a.problems, a.nodes, a.k1, mlp_dtype=a.mlp_dtype)
    dev = job.dev
    out = None
    for _ in range(3):
        out = job.run()
    torch.cuda.synchronize(dev)
    same = torch.equal(out, job.run_separate())
    walls = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(a.reps):
            job.run()
        torch.cuda.synchronize(dev)
        walls.append((time.perf_counter() - t0) / a.reps)
    wall = sorted(walls)[1]
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    tot = [0.0, 0.0, 0.0]
    for _ in range(a.reps):
        ev[0].record(); s = job.explore(); ev[1].record(); p = job.paths(s); ev[2].record(); job.smooth(p); ev[3].record()
        torch.cuda.synchronize(dev)
        for i in range(3):
            tot[i] += ev[i].elapsed_time(ev[i + 1])
    res = {'workload': 'BASELINE configs[4] as one job on one GPU: %d kuka14 problems, %d-node k1=%d RGGs (mean E %.0f), explorer '
                       '(loop 5, %s) -> paths of 5..35 waypoints from the scores -> 5 x smoother forward (d = 128, %s, loop 1), one '
                       'stream, no host sync in between; synthetic path rule and steering: see tools/cfg5_pipeline.py'
                       % (a.problems, a.nodes, a.k1, job.batch.total_edges / a.problems, a.mlp_dtype, a.mlp_dtype),
           'problems_per_s': round(a.problems / wall, 1), 'ms_per_job': round(wall * 1e3, 4),
           'timing': 'median of 3 blocks of %d back-to-back jobs: %s ms' % (a.reps, ' / '.join('%.4f' % (w * 1e3) for w in walls)),
           'stage_ms': {'explorer_forward': round(tot[0] / a.reps, 4), 'paths_from_scores (torch ops)': round(tot[1] / a.reps, 4),
                        'smoothing 5 x forward': round(tot[2] / a.reps, 4)},
           'waypoints_total': int(sum(job.counts)), 'pipelined_equals_separate_calls_bytewise': bool(same)}
    print(json.dumps(res))


if __name__ == '__main__':
    main()

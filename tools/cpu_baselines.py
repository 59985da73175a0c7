#!/usr/bin/env python
"""CPU baselines of the oracle (port of the reference's CPU forward) at the shapes of BASELINE.json's configs,
single-graph calls like eval_gnn.py:113-116, on THIS host (run it on the GPU box so the numbers sit beside the
GPU measurements).  Informational companion of bench.py's cpu_baseline field."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))
import torch  # noqa: E402
import gnnmp  # noqa: F401,E402
from gnnmp.weights import load_weights  # noqa: E402
from gnnmp.synth import ENVS, synth_graph  # noqa: E402
from oracle import ref_cpu  # noqa: E402


def measure(env, n, k1, threads, budget=6.0):
    w = load_weights(ENVS[env]['ckpt'])
    graphs = [synth_graph(env, n, k1, seed=1234 + i) for i in range(3)]
    torch.set_num_threads(threads)
    run = lambda g: ref_cpu.explorer_forward(w, g['v'], g['goal'], g['obstacles'], g['edge_index'], 5, materialize=True)  # noqa: E731
    run(graphs[0])
    cnt, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget and cnt < 64:
        run(graphs[cnt % 3])
        cnt += 1
    el = time.perf_counter() - t0
    return cnt / el, cnt, el


if __name__ == '__main__':
    allt = torch.get_num_threads()
    print('host threads available to torch: %d' % allt)
    for name, env, n, k1 in (('configs[0]', 'maze2', 200, 6), ('configs[1]', 'maze2', 1000, 8), ('configs[2] shape', 'kuka7', 2000, 10),
                             ('configs[4] shape', 'kuka14', 5000, 16), ('reference default (k=30)', 'maze2', 1002, 41)):
        for th in sorted({1, min(8, allt)}):
            r, c, el = measure(env, n, k1, th)
            print('%-26s %-7s N=%-5d k1=%-3d threads=%-3d %8.2f graphs/s  (%d calls in %.1f s)' % (name, env, n, k1, th, r, c, el))

#!/usr/bin/env python
"""Single-call latencies and other-config throughputs (informational; not the headline metric).
Run on the GPU box:  python tools/latency.py"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))
import torch  # noqa: E402
import gnnmp  # noqa: E402
from gnnmp.weights import load_weights  # noqa: E402
from gnnmp.synth import ENVS, synth_graph  # noqa: E402

dev = torch.device('cuda:0')


def model_for(env):
    e = ENVS[env]
    m = gnnmp.EncoderProcessDecoder(e['workspace'], e['C'], e['d'], e['S']).eval()
    m.load_state_dict(load_weights(e['ckpt']))
    return m, e


def timeit(fn, n=30, warm=5, reps=5):
    """Median over `reps` timed blocks of n calls (this pool shows sporadic multi-millisecond stalls that are unrelated to
    the kernels: round 1's "smoother bf16 C=14 batch 256 = 5.858 ms" line was one of them, tools/diag/smoother_c14.py)."""
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / n)
    return sorted(ts)[len(ts) // 2]


def single(env, n, k):
    m, e = model_for(env)
    g = {kk: (v.to(dev) if torch.is_tensor(v) else v) for kk, v in synth_graph(env, n, k).items()}
    dense = timeit(lambda: m(goal=g['goal'], loop=5, v=g['v'], obstacles=g['obstacles'], edge_index=g['edge_index']))
    sparse = timeit(lambda: m.edge_scores(g['goal'], 5, g['v'], g['obstacles'], g['edge_index']))
    b = m._single(g['goal'], g['v'], g['obstacles'], g['edge_index'])
    fb = timeit(lambda: m.forward_batch(b, 5))
    d2h = timeit(lambda: m(goal=g['goal'], loop=5, v=g['v'], obstacles=g['obstacles'], edge_index=g['edge_index']).cpu())
    graph, _ = m.capture(b, 5)
    rep = timeit(graph.replay)
    print('%-7s N=%-5d k1=%-3d E=%-7d single call: dense forward %.3f ms | sparse %.3f ms | prebuilt batch %.3f ms | '
          'dense + .cpu() %.3f ms | hipGraph replay %.3f ms' % (env, n, k, g['edge_index'].shape[1], dense * 1e3, sparse * 1e3,
                                                               fb * 1e3, d2h * 1e3, rep * 1e3))


def batched(env, n, k, G, uniq=16, dtype='fp32'):
    import gc
    gc.collect()
    torch.cuda.empty_cache()          # the previous configuration's workspace (GBs) must not be reclaimed inside the timing
    m, e = model_for(env)
    m.mlp_dtype = dtype
    base = [synth_graph(env, n, k, seed=1234 + i) for i in range(uniq)]
    b = gnnmp.GraphBatch.from_graphs([base[i % uniq] for i in range(G)], e['S'], dev)
    m.profile(dev, True)
    t = timeit(lambda: m.forward_batch(b, 5), n=10, warm=5)
    prof = m.profile_read(dev)
    print('%-7s N=%-5d k1=%-3d %s batch %-4d: %.3f ms/step = %.1f graphs/s   stages(ms/step): %s' %
          (env, n, k, dtype, G, t * 1e3, G / t, {kk: round(v[0] / max(v[1] // max(1, {'mp': 5}.get(kk, 1)), 1), 3) for kk, v in prof.items()}))


def smoother(name, C, P=20, F=500, Co=500, B=256, scale=1.0):
    """SURVEY.md 8(d) smoother addendum: ms per call (the reference calls it with loop = 1 five times per problem,
    smoother.py:243) and the batched rate."""
    import gc
    from gnnmp.planner import chain_edge_index
    from gnnmp.smoother import SmoothBatch
    gc.collect()
    torch.cuda.empty_cache()
    gen = torch.Generator().manual_seed(3)
    for dtype in ('fp32', 'bf16'):
        ms = gnnmp.ModelSmoother(workspace_size=3, config_size=C, embed_size=128, obs_size=6, scale=scale).eval()
        ms.load_state_dict(load_weights(name))
        ms.mlp_dtype = dtype
        mk = lambda n: (torch.rand(n, C, generator=gen) * 2 - 1)          # noqa: E731
        one = SmoothBatch([mk(P)], [mk(F)], [mk(Co)], [chain_edge_index(P)], dev)
        many = SmoothBatch([mk(P) for _ in range(B)], [mk(F) for _ in range(B)], [mk(Co) for _ in range(B)],
                           [chain_edge_index(P)] * B, dev)
        t1 = timeit(lambda: ms.forward_batch(one, 1))
        tb = timeit(lambda: ms.forward_batch(many, 1), n=10)
        print('%-18s C=%-2d %s  P=%d F=%d Co=%d: single call %.3f ms | batch of %d: %.3f ms = %.0f calls/s' %
              (name, C, dtype, P, F, Co, t1 * 1e3, B, tb * 1e3, B / tb))


if __name__ == '__main__':
  if '--smoother-only' not in sys.argv:
    single('maze2', 200, 6)        # BASELINE configs[0]
    single('maze2', 1000, 8)
    single('maze2', 1002, 41)      # the reference's default graph density (k=30 -> k1=41)
    single('kuka7', 2000, 10)
    batched('maze2', 200, 6, 256)
    batched('kuka7', 2000, 10, 64)       # configs[2] shape (fp32 here)
    batched('kuka14', 5000, 16, 32, uniq=4)     # configs[4] shape (fp32 here)
    batched('kuka7', 2000, 10, 64, dtype='bf16')       # configs[2]
    batched('kuka14', 5000, 16, 32, uniq=4, dtype='bf16')     # configs[4]
    batched('ur5', 1000, 8, 256)
    batched('snake7', 1000, 8, 256)
  if True:
    smoother('smooth_2d_attv3', 2)
    smoother('smooth_7d_attv3', 7)
    smoother('smooth_14d_attv3', 14)     # configs[4]: "+ smoother GNN"

#!/usr/bin/env python
"""Headline benchmark: RGG graphs/s of the GNN explorer forward on MI355X (BASELINE.json).

A step = one explorer forward (loop=5, use_obstacles=True, all E edge scores produced) over one
batch of 256 independent synthetic 1000-node k=8 maze2 RGGs (BASELINE.json configs[1]) with the
inputs already resident in HBM; everything the reference's forward() does per call is inside the
timed region (CSR build from the raw edge_index included).  N > 1: one process per GPU, every rank
owns its own 256 problems (weak scaling, no data-path collective); the only RCCL traffic is the
result gather after the timed region.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus 8 --steps 20 --warmup 5          # no launcher: bench.py starts its own 8 ranks (self_launch)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        --master-port P bench.py --gpus N --steps K --warmup W

`value` is timed with the per-stage HIP events switched OFF; the stage split and the roofline kernel's launch time come
from a second loop of the same K steps with the events on (`config.ms_per_step_with_stage_events`).  Every run also times
a STRONG-scaling leg (`config.strong_leg`: a fixed set of --strong-leg problems split over the ranks) next to the weak
headline, so one driver sweep over N = 1, 2, 4, 8 yields both curves.  At N = 1 the default command also times the two
bf16 shapes of BASELINE configs[2] / [4] (`config.other_configs_gpu`), the PCIe-inclusive, dense-output, bf16x3 and
single-graph legs, the planner leg (host loop on ONE core + device planner) and the CPU baseline; at N > 1 only what a
scaling line needs.  `config.whole_forward` separates CREDITED work (reference-formulation FLOPs / step time: not a
utilisation, may exceed 1) from EXECUTED work (counter-measured MFMA FLOPs per step, profiles/kernel_mfma.json).
torch's intra-op pool is capped at the container's CPU quota (gnnmp.hostenv): uncapped it gets the process throttled.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_FP32_TFLOPS = 157.3      # MI355X fp32 matrix = vector peak (MI355X_MICROARCH.md)
PEAK_BF16_TFLOPS = 2500.0     # dense bf16 MFMA peak
PEAK_HBM_GBS = 8000.0


def algorithmic_flops(N, E, O, C, d, S, L):
    """FLOPs of the reference formulation per graph (2 per MAC)."""
    f_enc = 2 * N * (4 * C * d + d * d) + 2 * 2 * E * (2 * C * d + d * d) + 2 * N * (C * d + d * d) \
        + 2 * 2 * O * (S * d + d * d)
    f_att = 3 * ((N + E) * (10 * d * d + 4 * d * (O + 1)) + 16 * O * d * d)
    f_loop = L * (8 * N * d * d + 12 * E * d * d + E * d + 4 * N * d * d) + 4 * N * d * d
    f_pol = E * (8 * d * d + 2 * d)
    return f_enc + f_att + f_loop + f_pol


def edge_pre_flops(E, O, C, d):
    """Reference-formulation FLOPs of what the dominant kernel (edge_pre) replaces: the two edge
    encoders (model.py:120,123) and the three edge attention blocks (model.py:128-130, map rows)."""
    return 2 * 2 * E * (2 * C * d + d * d) + 3 * E * (10 * d * d + 4 * d * (O + 1))


def mp_fused_bytes(N, E, d, bf16):
    """HBM bytes one mp_fused launch (one message-passing iteration, model.py:139-143) has to move for one graph: the
    per-edge first-layer constant K_e and the packed edge record stream in once; every node row of A (gathered by source:
    the gather itself is served by L2, its first touch is HBM), X and R is read once and X', A' are written once.  (Until
    late in round 3 a tile's B' rows were also written and read back, N d se each way; they are now recomputed from the
    tile's X rows -- the count below is the smaller, current one.)  bf16 mode stores K_e, A and X in bf16."""
    se = 2 if bf16 else 4
    return E * (d * se + 4) + N * d * (2 * se + 4) + N * d * 2 * se


def node_side_flops(N, O, C, d, S):
    """Reference-formulation FLOPs of the node side (obs + node_pre stages): node encoders (model.py:119,122), the obstacle
    encoders and the K / V projections of the obstacle codes for all six blocks (model.py:125-130), three node attention
    blocks."""
    return 2 * N * (4 * C * d + d * d) + 2 * N * (C * d + d * d) + 2 * 2 * O * (S * d + d * d) + 3 * 16 * O * d * d \
        + 3 * N * (10 * d * d + 4 * d * (O + 1))


def policy_bytes(N, E, d, bf16):
    """HBM bytes of the policy launch for one graph: the per-edge first-layer constant PE_e and the CSR record (16 B) in, one
    score out; the PS / PT rows gathered by source / target are served by L2 after their first touch."""
    se = 2 if bf16 else 4
    return E * (d * se + 16 + 4) + 2 * N * d * se


def algorithmic_bytes(N, E, O, C, S):
    """SURVEY.md section 8(d) B_sparse: v, goal, obstacles, int32 edge pairs, scores out."""
    return 4 * N * C + 4 * C + 4 * O * S + 8 * E + 4 * E


def kernel_source_hash():
    """sha256 over the sources of the dominant kernel (what the entries of profiles/kernel_traffic.json were measured on)."""
    import hashlib
    hsh = hashlib.sha256()
    for f in ('chain.hpp', 'layout.hpp', 'kernels.hpp', 'explorer_kernels.hip'):
        with open(os.path.join(REPO, 'gnn-motion-planning_amd', 'csrc', f), 'rb') as fh:
            hsh.update(fh.read())
    return hsh.hexdigest()


def cpu_baseline(env, n_nodes, k1, budget_s, seed0):
    """The oracle (a port of the reference's CPU path, materialising attention form) timed on this
    host: single-graph calls in a loop exactly like eval_gnn.py:113-116,194."""
    from gnnmp.weights import load_weights
    from gnnmp.synth import ENVS, synth_graph
    from oracle import ref_cpu
    w = load_weights(ENVS[env]['ckpt'])
    graphs = [synth_graph(env, n_nodes, k1, seed=seed0 + i) for i in range(4)]
    run = lambda g: ref_cpu.explorer_forward(w, g['v'], g['goal'], g['obstacles'], g['edge_index'], 5,  # noqa: E731
                                             materialize=True)
    from gnnmp.hostenv import cpu_quota
    all_threads = min(torch.get_num_threads(), cpu_quota())         # what this container may really use (cgroup quota), not the machine's cores
    results = []
    for threads in sorted({min(8, all_threads), all_threads}):      # 8 = the survey container's count; all = this host
        torch.set_num_threads(threads)
        run(graphs[0])                  # warm-up
        n, t0 = 0, time.perf_counter()
        while True:
            run(graphs[n % len(graphs)])
            n += 1
            el = time.perf_counter() - t0
            if el > budget_s / 2 or n >= 64:
                break
        results.append((n / el, threads, n, el))
    # the other BASELINE config shapes, single-graph calls like the reference makes them:
    # cfg 1 (maze2 200-node k=6: the reference's own CPU-runnable case), cfg 3 (kuka7 2000-node k=10), cfg 5 (kuka14
    # 5000-node k=16); a few calls each at the better thread count
    torch.set_num_threads(max(results)[1])
    others = {}
    for name, (oenv, on, ok, calls) in {'cfg1_maze2_N200_k6': ('maze2', 200, 6, 8), 'cfg3_kuka7_N2000_k10': ('kuka7', 2000, 10, 2),
                                        'cfg5_kuka14_N5000_k16': ('kuka14', 5000, 16, 1)}.items():
        ow = load_weights(ENVS[oenv]['ckpt'])
        og = synth_graph(oenv, on, ok, seed=seed0)
        orun = lambda: ref_cpu.explorer_forward(ow, og['v'], og['goal'], og['obstacles'], og['edge_index'], 5, materialize=True)  # noqa: E731
        orun()
        t0 = time.perf_counter()
        for _ in range(calls):
            orun()
        others[name] = round(calls / (time.perf_counter() - t0), 3)
    torch.set_num_threads(all_threads)
    best = max(results)
    return {'value': best[0], 'unit': 'graphs/s', 'cores': best[1], 'kind': 'port',
            'sample': 'single-graph oracle forwards (%s N=%d k1=%d loop=5, attention materialised as model.py:178-179): '
                      % (env, n_nodes, k1) + '; '.join('%d calls in %.1f s on %d torch threads = %.2f graphs/s' %
                                                        (r[2], r[3], r[1], r[0]) for r in results) + '; best reported',
            'other_configs_graphs_per_s': others}


def roofline_blocks(w, e, G, prof, Ns, Es, Os):
    """(`roofline` block of the dominant kernel, every stage against the roof that bounds it) of one workload `w` (env, nodes, k1,
    mlp_dtype, loop) from the per-stage HIP-event times `prof` of its profiled loop.  Work counts: SURVEY.md section 8(d)."""
    is_bf16 = w.mlp_dtype == 'bf16'
    ep_flops = sum(edge_pre_flops(m, o, e['C'], e['d']) for m, o in zip(Es, Os))
    ep_ms, ep_n = prof['edge_pre']
    ep_avg_ms = ep_ms / max(ep_n, 1)
    achieved = ep_flops / (ep_avg_ms * 1e-3) / 1e12 if ep_avg_ms > 0 else 0.0
    # the roofline block describes the DOMINANT kernel of this workload: the stage with the largest share of the step
    stage_tot = {k: v[0] for k, v in prof.items()}
    dom = max(stage_tot, key=stage_tot.get)
    if dom not in ('edge_pre', 'mp'):
        dom = 'edge_pre'
    # bf16 shapes where the two stages are within 3 % of each other (configs[4] shape: 0.474 / 0.476 ms): keep the block on the
    # message-passing kernel, so that it does not flip between runs (the edge stage is in stage_roofline either way)
    if w.mlp_dtype == 'bf16' and dom == 'edge_pre' and stage_tot.get('mp', 0.0) >= 0.97 * stage_tot['edge_pre']:
        dom = 'mp'
    pname = {'fp32': '0', 'bf16': '1', 'bf16x3': '2'}[w.mlp_dtype]
    if dom == 'mp':
        mp_ms, mp_n = prof['mp']
        mp_avg_ms = mp_ms / max(mp_n, 1)
        mp_bytes = sum(mp_fused_bytes(n, m, e['d'], w.mlp_dtype == 'bf16') for n, m in zip(Ns, Es))
        # d = 64 runs the eight-wave form (mp_fused_w8_kernel, round 5) in the fp32 and bf16 modes
        mp_name = 'mp_fused_w8_kernel<64, %s' % pname if (e['d'] == 64 and pname in ('0', '1')) else 'mp_fused_kernel<%d, %s' % (e['d'], pname)
        roof = {'kernel': '%s...> (one message-passing iteration: edge MLP second layer, max aggregation, '
                          'node update; %d launches per step)' % (mp_name, w.loop),
                'kernel_like': mp_name,
                'bound': 'hbm', 'achieved': round(mp_bytes / (mp_avg_ms * 1e-3) / 1e9, 1) if mp_avg_ms > 0 else 0.0,
                'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'launch_ms': round(mp_avg_ms, 4), 'algorithmic_bytes_per_launch': mp_bytes}
    else:
        peak = PEAK_BF16_TFLOPS if w.mlp_dtype == 'bf16' else PEAK_FP32_TFLOPS
        roof = {'kernel': '%s<%d, %s, EDGE> (edge encoders + 3 obstacle-attention blocks)' % ('pre_resident_kernel' if (pname == '1' or (pname == '0' and e['d'] == 32)) else 'pre_kernel', e['d'], pname),
                'kernel_like': '%s<%d, %s, true' % ('pre_resident_kernel' if (pname == '1' or (pname == '0' and e['d'] == 32)) else 'pre_kernel', e['d'], pname),
                'bound': 'mfma', 'achieved': round(achieved, 2), 'peak': peak, 'unit': 'TFLOP/s',
                'launch_ms': round(ep_avg_ms, 4), 'algorithmic_flops_per_launch': ep_flops}
    roof['frac'] = round(roof['achieved'] / roof['peak'], 4)
    # every stage against the roof that bounds it (the `roofline` block above is the dominant one of these)
    is_bf16 = w.mlp_dtype == 'bf16'
    mfma_peak = PEAK_BF16_TFLOPS if is_bf16 else PEAK_FP32_TFLOPS
    def per_launch(stage):
        ms, n = prof.get(stage, (0.0, 0))
        return ms / max(n, 1)
    stage_roof = {}
    ns_ms = per_launch('obs') + per_launch('node_pre')
    ns_flops = sum(node_side_flops(n, o, e['C'], e['d'], e['S']) for n, o in zip(Ns, Os))
    for name, ms, work, bound in (
            ('edge_pre', ep_avg_ms, ep_flops, 'mfma'), ('obs+node_pre', ns_ms, ns_flops, 'mfma'),
            ('mp (per launch)', per_launch('mp'), sum(mp_fused_bytes(n, m, e['d'], is_bf16) for n, m in zip(Ns, Es)), 'hbm'),
            ('policy', per_launch('policy'), sum(policy_bytes(n, m, e['d'], is_bf16) for n, m in zip(Ns, Es)), 'hbm')):
        if ms <= 0:
            continue
        if bound == 'mfma':
            ach = work / (ms * 1e-3) / 1e12
            stage_roof[name] = {'bound': 'mfma', 'ms': round(ms, 4), 'TFLOPs': round(ach, 1), 'frac': round(ach / mfma_peak, 3)}
        else:
            ach = work / (ms * 1e-3) / 1e9
            stage_roof[name] = {'bound': 'hbm', 'ms': round(ms, 4), 'GBs': round(ach, 0), 'frac': round(ach / PEAK_HBM_GBS, 3)}
            if name.startswith('mp'):
                # the same launch against the matrix pipe: 2 d^2 FLOP per edge (message layer) + 10 d^2 per padded node row (W_dst X,
                # W_lx X, W_la agg, M1 H, M2 Y; the last iteration's M3 is not counted).  At d = 64 in fp32 the two roofs meet
                # (16 FLOP per algorithmic byte against a ridge of 19.7): the launch is priced against both
                d_ = e['d']
                fl = sum(2.0 * d_ * d_ * m + 5 * 2.0 * d_ * d_ * (((n + 31) // 32) * 32) for n, m in zip(Ns, Es))
                tf = fl / (ms * 1e-3) / 1e12
                stage_roof[name].update({'TFLOPs': round(tf, 1), 'frac_mfma': round(tf / mfma_peak, 3)})
    # issue-slot fraction of the edge pre kernel (bf16 mode: it is bound by instruction issue, not by the matrix pipe its FLOPs are
    # priced against): (4 x SQ_ACTIVE_INST_VALU + SQ_VALU_MFMA_BUSY_CYCLES) / SIMD cycles from separate --pmc passes
    # (tools/issue_json.py -> profiles/kernel_issue.json, stamped with workload and source hash like the traffic entries)
    ipath = os.path.join(REPO, 'profiles', 'kernel_issue.json')
    if os.path.exists(ipath) and 'edge_pre' in stage_roof:
        try:
            wk_ = '%s N=%d k1=%d graphs=%d %s' % (w.env, w.nodes, w.k1, G, w.mlp_dtype)
            for ij in json.load(open(ipath)):
                if ij.get('workload') == wk_ and ij.get('stage') == 'edge_pre':
                    stage_roof['edge_pre']['issue_slots'] = {k_: ij.get(k_) for k_ in (
                        'issue_slot_frac', 'valu_frac', 'mfma_frac', 'valu_per_32_row_tile', 'mfma_per_32_row_tile', 'measured')}
                    stage_roof['edge_pre']['issue_slots']['stale'] = ij.get('kernel_source_sha256') != kernel_source_hash()
        except Exception:
            pass
    # HBM bytes per launch of that kernel come from separate rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE cannot be
    # read inside this process; tools/traffic_json.py writes profiles/kernel_traffic.json): every entry carries the
    # workload it was measured on, the date of the pass and a hash of the kernel sources; the number is reported only
    # for the same workload and while that hash still matches the sources of the library in use
    wkey = '%s N=%d k1=%d graphs=%d %s' % (w.env, w.nodes, w.k1, G, w.mlp_dtype)
    roof['traffic'] = None
    tpath = os.path.join(REPO, 'profiles', 'kernel_traffic.json')
    if os.path.exists(tpath):
        try:
            for tj in json.load(open(tpath)):
                if tj.get('kernel_like') == roof['kernel_like'] and tj.get('workload') == wkey:
                    fresh = tj.get('kernel_source_sha256') == kernel_source_hash()
                    roof['traffic'] = tj.get('hbm_bytes_per_launch') if fresh else None
                    roof['traffic_source'] = {'file': 'profiles/kernel_traffic.json', 'measured': tj.get('measured'),
                                              'kernel_source_sha256': tj.get('kernel_source_sha256', '')[:16],
                                              'stale': not fresh, 'how': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, '
                                              'gfx950 corrections of MI355X_MICROARCH.md'}
        except Exception:
            pass
    # launch time of the roofline kernel as the committed rocprofv3 --kernel-trace run saw it (timed launches only, warm-ups
    # dropped: tools/rocprof_summary.py --warmup / --steps -> profiles/kernel_launch_ms.json), next to the in-process HIP-event
    # figure `launch_ms`; the profiler's own overhead sits in the gap between the two
    lpath = os.path.join(REPO, 'profiles', 'kernel_launch_ms.json')
    if os.path.exists(lpath):
        try:
            for lj in json.load(open(lpath)):
                if lj.get('kernel_like') == roof['kernel_like'] and lj.get('workload') == wkey:
                    roof['launch_ms_rocprof'] = lj.get('avg_ms_timed_launches')
                    work = roof.get('algorithmic_flops_per_launch', roof.get('algorithmic_bytes_per_launch'))
                    unit = 1e12 if roof['bound'] == 'mfma' else 1e9
                    roof['frac_rocprof'] = round(work / (lj['avg_ms_timed_launches'] * 1e-3) / unit / roof['peak'], 4)
                    roof['launch_ms_rocprof_source'] = {'file': 'profiles/kernel_launch_ms.json', 'measured': lj.get('measured'),
                                                        'launches': lj.get('launches'), 'stale': lj.get('kernel_source_sha256') != kernel_source_hash()}
        except Exception:
            pass
    return roof, stage_roof


def executed_mfma(wkey):
    """Matrix-pipe FLOPs one step of workload `wkey` EXECUTES (padding rows, recomputation and all), from profiles/kernel_mfma.json:
    per kernel, SQ_INSTS_VALU_MFMA_MOPS_* x 512 (or SQ_INSTS_MFMA x the FLOPs of the kernel's MFMA shape) per launch x its
    launches per step, written by tools/mfma_json.py from separate rocprofv3 --pmc passes and stamped with the kernel source hash
    like the traffic entries.  None when there is no entry for this workload."""
    path = os.path.join(REPO, 'profiles', 'kernel_mfma.json')
    if not os.path.exists(path):
        return None
    try:
        for ent in json.load(open(path)):
            if ent.get('workload') == wkey:
                return {'flops_per_step': float(ent['executed_mfma_flops_per_step']), 'measured': ent.get('measured'),
                        'stale': ent.get('kernel_source_sha256') != kernel_source_hash(), 'how': ent.get('how')}
    except Exception:
        pass
    return None


def whole_forward_block(wkey, flops_step, bytes_step, step_s, is_bf16):
    """The whole forward against the peaks of one GPU.  `credited_*`: FLOPs of the REFERENCE formulation (SURVEY.md section 8(d):
    what the reference would have computed for these graphs) over the step time -- the kernels reach the same results with fewer
    operations (algebraic rewrites), so a credited fraction can exceed what the matrix pipe executed and even 1.0;
    `executed_*`: the MFMA FLOPs the kernels really issue per step (counter-measured, executed_mfma) over the same time."""
    peak = PEAK_BF16_TFLOPS if is_bf16 else PEAK_FP32_TFLOPS
    peak_name = 'bf16_mfma_peak' if is_bf16 else 'fp32_peak'
    out = {'credited_TFLOPs': round(flops_step / step_s / 1e12, 2), 'credited_frac_' + peak_name: round(flops_step / step_s / 1e12 / peak, 4),
           'credited': 'reference-formulation FLOPs (SURVEY.md 8(d)) / step time; not a utilisation figure -- see executed_*',
           'algorithmic_GBs': round(bytes_step / step_s / 1e9, 3), 'frac_hbm_peak': round(bytes_step / step_s / 1e9 / PEAK_HBM_GBS, 6),
           'executed_TFLOPs': None, 'frac_executed': None}
    ex = executed_mfma(wkey)
    if ex is not None:
        out.update({'executed_TFLOPs': round(ex['flops_per_step'] / step_s / 1e12, 2),
                    'frac_executed': round(ex['flops_per_step'] / step_s / 1e12 / peak, 4),
                    'executed_source': {'file': 'profiles/kernel_mfma.json', 'measured': ex['measured'], 'stale': ex['stale'],
                                        'mfma_flops_per_step': ex['flops_per_step'], 'how': ex['how']}})
    return out


OTHER_CONFIGS = (      # (key, env, nodes, k1, problems per batch, MFMA operand mode): the bf16 shapes of BASELINE configs[2] and [4] (str2name.py:46-64)
    ('cfg2_kuka7_N2000_k10_bf16', 'kuka7', 2000, 10, 64, 'bf16'),
    ('cfg4_kuka14_N5000_k16_bf16', 'kuka14', 5000, 16, 32, 'bf16'))


def other_config_leg(key, env, nodes, k1, G, mlp_dtype, loop, steps, warmup, dev):
    """One of the OTHER BASELINE shapes under the same clock as the headline: `warmup` + `steps` forwards timed like `value`
    (synchronize both sides, per-stage events off), then the same `steps` once more with the events on for the stage split and
    the dominant kernel's roofline block (same rules as the headline's: roofline_blocks)."""
    import gnnmp
    from gnnmp.weights import load_weights
    from gnnmp.synth import ENVS, synth_batch_gpu
    e = ENVS[env]
    model = gnnmp.EncoderProcessDecoder(e['workspace'], e['C'], e['d'], e['S']).eval()
    model.load_state_dict(load_weights(e['ckpt']), strict=True)
    model.mlp_dtype = mlp_dtype
    graphs = synth_batch_gpu(env, nodes, k1, G, dev, seed0=1234)
    batch = gnnmp.GraphBatch.from_graphs(graphs, e['S'], dev)

    def loop_of(n):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(n):
            out = model.forward_batch(batch, loop)
        torch.cuda.synchronize(dev)
        return time.perf_counter() - t0, out
    loop_of(warmup)
    el, scores = loop_of(steps)
    model.profile(dev, True)
    loop_of(1)
    model.profile_read(dev)                                # drop the warm-up step of the profiled loop
    loop_of(steps)
    prof = model.profile_read(dev)
    model.profile(dev, False)
    Ns = [int(g['v'].shape[0]) for g in graphs]
    Es = [int(g['edge_index'].shape[1]) for g in graphs]
    Os = [g['obstacles'].reshape(-1, e['S']).shape[0] for g in graphs]
    w = argparse.Namespace(env=env, nodes=nodes, k1=k1, mlp_dtype=mlp_dtype, loop=loop)
    roof, stage_roof = roofline_blocks(w, e, G, prof, Ns, Es, Os)
    wkey = '%s N=%d k1=%d graphs=%d %s' % (env, nodes, k1, G, mlp_dtype)
    flops = sum(algorithmic_flops(n, m, o, e['C'], e['d'], e['S'], loop) for n, m, o in zip(Ns, Es, Os))
    nbytes = sum(algorithmic_bytes(n, m, o, e['C'], e['S']) for n, m, o in zip(Ns, Es, Os))
    res = {'workload': '%s: batch of %d problems, %d-node k1=%d RGGs (mean E=%.0f, O=%d), loop=%d, real %s checkpoint, %s operands'
                       % (env, G, nodes, k1, sum(Es) / len(Es), Os[0], loop, e['ckpt'], mlp_dtype),
           'graphs_per_s': round(G * steps / el, 1), 'ms_per_step': round(el / steps * 1e3, 4), 'steps': steps, 'warmup': warmup,
           'dtype': mlp_dtype, 'stage_ms_per_step': {k_: round(v_[0] / max(steps, 1), 4) for k_, v_ in prof.items()},
           'roofline': {k_: roof.get(k_) for k_ in ('kernel_like', 'bound', 'achieved', 'peak', 'unit', 'frac', 'launch_ms', 'traffic',
                                                    'algorithmic_bytes_per_launch', 'algorithmic_flops_per_launch') if roof.get(k_) is not None or k_ == 'traffic'},
           'traffic_stale': (roof.get('traffic_source') or {}).get('stale'),
           'stage_frac': {k_: (v_['frac'], v_['bound']) for k_, v_ in stage_roof.items()},
           'whole_forward': whole_forward_block(wkey, flops, nbytes, el / steps, mlp_dtype == 'bf16'),
           'result_checksum': float(scores.double().sum().item())}
    del model, batch, graphs, scores
    torch.cuda.empty_cache()
    return res


def planner_leg(n_host, n_device, dev):
    """north_star: "collision checks and the sequential planner control flow stay on the host CPU and are timed in the
    same run (core count stated)".  Reference split first -- GNN forwards on the GPU, sampling + greedy loop + every
    collision check + steering on ONE host core (planner.explore, the eval_gnn.py:168-276 counterpart) -- then the same
    problems with the planner itself on the device (planner.explore_maze_batch).  2-D maze problems of the reference's
    published run (mazes_hard.npz, batch = t_max = 500, k = 30, smoothing on)."""
    import numpy as np
    import gnnmp
    from gnnmp import planner
    from gnnmp.maze2d import Maze2D
    from gnnmp.weights import load_weights
    with np.load(os.path.join(REPO, 'tests', 'golden', 'evalset_mazehard_first1000.npz')) as f:
        env = Maze2D(f['maps'], f['init_states'], f['goal_states'])
    m = gnnmp.EncoderProcessDecoder(2, 2, 32, 2).eval()
    m.load_state_dict(load_weights('weights_maze'))
    ms = gnnmp.ModelSmoother(workspace_size=2, config_size=2, embed_size=128, obs_size=6).eval()
    ms.load_state_dict(load_weights('smooth_2d_attv3'))
    # `host_cores: 1` is enforced, not assumed: torch's intra-op pool (sized from the machine, not from the container's CPU quota)
    # would otherwise spin on every core it can get between the loop's small parallel ops and get the process throttled
    # (gnnmp/hostenv.py; the 0.86 -> 5.0 ms `gnn_forward_ms_per_problem` of BENCH_r05 was exactly that)
    threads0 = torch.get_num_threads()
    torch.set_num_threads(1)
    np.random.seed(1234)
    for i in range(3):                                                                           # warm-up (first use of every kernel variant)
        env.init_new_problem(i)
        planner.explore(env, m, ms, True, batch=500, t_max=500, k=30, device=dev)
    fwd = tot = 0.0
    checks = 0
    split = {}
    t0 = time.perf_counter()
    for i in range(n_host):
        env.init_new_problem(i)
        r = planner.explore(env, m, ms, True, batch=500, t_max=500, k=30, device=dev)
        fwd += r['forward']; tot += r['total']; checks += r['c_explore'] + r['c_smooth']
        for k_, v_ in r['forward_split'].items():
            split[k_] = split.get(k_, 0) + v_
    wall_host = time.perf_counter() - t0
    # the device planner is quoted at ONE size everywhere (README, DESIGN, this line): n_device problems (the evaluation set
    # cycled), median of three timed passes after a warm-up pass at the same size (allocator, kernel variants)
    idx = [i % len(env.maps) for i in range(n_device)]
    planner.eval_gnn_device(env, idx, m, ms, device=dev)                                          # warm-up
    walls = []
    for _ in range(3):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        out = planner.eval_gnn_device(env, idx, m, ms, device=dev)
        torch.cuda.synchronize(dev)
        walls.append(time.perf_counter() - t0)
    wall_dev = sorted(walls)[1]
    torch.set_num_threads(threads0)
    return {'problems': 'mazes_hard.npz (2-D maze), batch = t_max = 500, k = 30, smoothing on',
            'host_loop': {'problems': n_host, 'problems_per_s': round(n_host / wall_host, 2), 'host_cores': 1,
                          'gnn_forward_ms_per_problem': round(1e3 * fwd / n_host, 2),
                          # what that span (eval_gnn.py:193-196) is made of, ms per problem: obs_data = host tensors of the samples +
                          # their H2D copies, h2d = graph tensors, module_call = host time of model(**kw) (enqueue only),
                          # d2h_wait = .cpu() of the dense [N, N] block (kernels finishing + the 4 MB pageable copy)
                          'gnn_forward_split_ms': {k_: round(1e3 * v_ / n_host, 3) for k_, v_ in split.items() if k_ != 'calls'},
                          'gnn_forwards_per_problem': round(split.get('calls', 0) / n_host, 2),
                          'host_ms_per_problem': round(1e3 * (tot - fwd) / n_host, 2),
                          'collision_checks_per_problem': round(checks / n_host, 1),
                          'what': 'dense drop-in forward on the GPU; sampling, greedy loop, collision checks, steering on one host core'},
            'device_planner': {'problems': n_device, 'problems_per_s': round(n_device / wall_dev, 1), 'host_cores': 3,
                               'timing': 'median of 3 passes over the same %d problems: %s problems/s' % (
                                   n_device, ' / '.join('%.0f' % (n_device / w) for w in walls)),
                               'success': int(out[0]), 'collision_checks_per_problem': round(out[1], 2),
                               'what': 'sampling on one host thread, two more host threads drive device passes of 128 problems on their '
                                       'own streams; graphs, forwards, greedy loop, collision checks, steering on the GPU'}}


def self_launch(n_gpus):
    """`python bench.py --gpus N` with N > 1 and no launcher environment (RANK / WORLD_SIZE unset): start the N ranks ourselves
    -- the same `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
    bench.py <same flags>` the driver would have used, on a free port -- let rank 0's one JSON line through on stdout and
    return the launcher's exit status (a failing rank's traceback arrives on stderr through the inherited stream, and
    torch.distributed.run names the failing rank).  What is sharded and what is gathered: eval_gnn.py:113-122."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')        # dmabuf IPC only on this driver (RCCL needs it across processes)
    env.setdefault('OMP_NUM_THREADS', '8')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n_gpus), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print('[bench] no launcher environment: starting %d ranks: %s' % (n_gpus, ' '.join(cmd)), file=sys.stderr, flush=True)
    rc = subprocess.call(cmd, env=env)
    if rc != 0:
        print('[bench] the %d-rank run failed with exit status %d (the failing rank is named above)' % (n_gpus, rc), file=sys.stderr, flush=True)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--graphs', type=int, default=256, help='graphs per GPU per step')
    ap.add_argument('--strong', type=int, default=0, metavar='N_TOTAL',
                    help='strong scaling: a FIXED set of N_TOTAL problems (seeds 1234 .. 1234 + N_TOTAL - 1) split over the '
                         'ranks by gnnmp.dist.shard_range; --graphs is ignored.  Default (0): weak scaling, --graphs per GPU')
    ap.add_argument('--gather-reps', type=int, default=5, help='timed repetitions of the final result gather (median reported)')
    ap.add_argument('--env', default='maze2')
    ap.add_argument('--nodes', type=int, default=1000)
    ap.add_argument('--k1', type=int, default=8)
    ap.add_argument('--loop', type=int, default=5)
    ap.add_argument('--mlp-dtype', default='fp32', choices=['fp32', 'bf16', 'bf16x3'],
                    help="MFMA operand precision; the headline metric is fp32 (bf16 = BASELINE configs[2]/[4] mode)")
    ap.add_argument('--cpu-seconds', type=float, default=15.0)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--pcie-steps', type=int, default=20, help='extra steps timed incl. H2D/D2H (0 = skip)')
    ap.add_argument('--dense-steps', type=int, default=5, help='extra steps timed with the dense N x N output (0 = skip)')
    ap.add_argument('--inflight-steps', type=int, default=0,
                    help='extra steps timed with TWO batches in flight on two HIP streams (own workspaces and score buffers): what a '
                         'serving loop over independent batches gets when one forward fills the launch tails of the other; reported next '
                         'to, never instead of, `value`.  Off by default (0): its overlapped launches would enter the per-kernel '
                         'averages of a `rocprofv3 --stats` run of the default command; tools/profile_r06.sh runs it with 20')
    ap.add_argument('--inflight-depth', type=int, default=2, help='batches in flight for --inflight-steps (streams, workspaces, score buffers)')
    ap.add_argument('--bf16x3-steps', type=int, default=5,
                    help='extra steps timed in the opt-in bf16x3 mode (fp32-class results from the bf16 matrix pipe, '
                         'DESIGN.md 4.2b); reported next to, never instead of, `value`; only with --mlp-dtype fp32')
    ap.add_argument('--single-steps', type=int, default=50, help='single-graph latency: calls per timed block (0 = skip)')
    ap.add_argument('--unique', type=int, default=0, help='distinct synthetic graphs per GPU (0 = all)')
    ap.add_argument('--planner-problems', type=int, default=16,
                    help='host-loop planner problems timed next to the forward benchmark (0 = skip the planner leg)')
    ap.add_argument('--strong-leg', type=int, default=1024, metavar='N_TOTAL',
                    help='also time a strong-scaling leg in the same run: the FIXED problem set 0 .. N_TOTAL - 1 split over the ranks, '
                         'reported as config.strong_leg next to the weak headline (0 = skip; ignored with --strong)')
    ap.add_argument('--strong-steps', type=int, default=5, help='timed steps of the strong leg')
    ap.add_argument('--other-configs-steps', type=int, default=20,
                    help='timed steps of each of the other BASELINE shapes (configs[2] kuka7 2000 x 64 bf16, configs[4] kuka14 5000 x 32 bf16) '
                         'run after the headline workload and reported as config.other_configs_gpu; only with the default workload on one GPU (0 = skip)')
    ap.add_argument('--launch-check', action='store_true',
                    help='start the ranks, connect them (init_process_group + one all_reduce), print one JSON line and exit without '
                         'running the workload (tests the launcher on a box without GPUs: GNNMP_BENCH_BACKEND=gloo)')
    args = ap.parse_args()

    have_launcher = 'RANK' in os.environ and 'WORLD_SIZE' in os.environ
    if args.gpus > 1 and not have_launcher:
        # the driver's N = 1 command shape with a larger N: be our own launcher instead of failing
        raise SystemExit(self_launch(args.gpus))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d -- either launch one rank per GPU (python -m torch.distributed.run '
                         '--nnodes=1 --nproc-per-node %d --master-addr 127.0.0.1 --master-port P bench.py --gpus %d ...) or run '
                         '`python bench.py --gpus %d` WITHOUT RANK / WORLD_SIZE in the environment and it starts the ranks itself'
                         % (args.gpus, world, args.gpus, args.gpus, args.gpus))
    # GNNMP_BENCH_BACKEND=gloo: collectives over gloo on host tensors (launcher tests on a box without GPUs; two ranks sharing
    # one GPU, which RCCL refuses).  Default nccl = RCCL over xGMI on device tensors.
    backend = os.environ.get('GNNMP_BENCH_BACKEND', 'nccl')
    if backend not in ('nccl', 'gloo'):
        raise SystemExit('bench.py: GNNMP_BENCH_BACKEND must be nccl or gloo, got %r' % backend)
    if world > 1:
        # a scaling line needs the warm-up, `value`, the strong leg, the profiled loop and the gather -- not N copies of the PCIe,
        # dense-output, bf16x3, in-flight and planner legs; rank 0 alone keeps the single-graph latency
        args.pcie_steps = args.dense_steps = args.bf16x3_steps = args.inflight_steps = args.other_configs_steps = 0
        if rank != 0:
            args.single_steps = 0
    if args.strong > 0 and args.strong < world:
        # every rank sees the same arguments and leaves together (a rank bailing out alone would leave the others in a collective)
        raise SystemExit('bench.py: --strong %d is fewer problems than ranks (%d)' % (args.strong, world))
    if args.launch_check:
        if backend == 'nccl':
            torch.cuda.set_device(local)
        dist.init_process_group(backend, **({'device_id': torch.device('cuda', local)} if backend == 'nccl' else {}))
        ones = torch.ones(1, dtype=torch.float64, device=torch.device('cuda', local) if backend == 'nccl' else 'cpu')
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        if rank == 0:
            print(json.dumps({'launch_check': True, 'n_gpus': world, 'ranks_seen': int(round(float(ones.item()))), 'backend': backend,
                              'self_launched': os.environ.get('TORCHELASTIC_RUN_ID') is not None}), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return
    n_dev = torch.cuda.device_count()
    if n_dev < 1:
        raise SystemExit('bench.py: no GPU visible (the forward has no CPU path)')
    if backend == 'nccl' and local >= n_dev:
        raise SystemExit('bench.py: LOCAL_RANK %d but only %d GPU(s) visible' % (local, n_dev))
    torch.cuda.set_device(local % n_dev)
    dev = torch.device('cuda', local % n_dev)
    # GNNMP_BENCH_FORCE_DIST=1 exercises the RCCL code path (init, barrier, all_reduce, all_gather) with one rank
    use_dist = world > 1 or os.environ.get('GNNMP_BENCH_FORCE_DIST') == '1'
    cdev = dev if backend == 'nccl' else torch.device('cpu')       # where the collectives' tensors live
    if use_dist:
        dist.init_process_group(backend, **({'device_id': dev} if backend == 'nccl' else {}))

    import gnnmp
    from gnnmp.weights import load_weights
    from gnnmp.synth import ENVS, synth_batch_gpu
    from gnnmp.hostenv import cpu_quota, limit_host_threads
    machine_threads = torch.get_num_threads()
    # the container's CPU quota shared by the ranks of this node (16 CPUs on the MI355X job boxes, where torch would start 128 threads)
    host_threads = limit_host_threads(max(1, cpu_quota() // max(world, 1)))
    e = ENVS[args.env]
    model = gnnmp.EncoderProcessDecoder(e['workspace'], e['C'], e['d'], e['S']).eval()
    model.load_state_dict(load_weights(e['ckpt']), strict=True)
    model.mlp_dtype = args.mlp_dtype

    def sync():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def problem_set(G, seed0, unique=0):
        """G problems, problem i = synth_graph(seed = seed0 + i): same node / obstacle draws as the host generator; the kNN
        edge lists come from the device graph builder (bit-identical to the host builder), so start-up takes seconds."""
        uniq = G if unique <= 0 else min(G, unique)
        base = synth_batch_gpu(args.env, args.nodes, args.k1, uniq, dev, seed0=seed0)
        graphs = [base[i % uniq] for i in range(G)]
        return graphs, gnnmp.GraphBatch.from_graphs(graphs, e['S'], dev)

    def timed(batch, warmup, steps):
        """W untimed steps, then exactly K steps between barrier + synchronize on both sides; this rank's seconds."""
        out = None
        for _ in range(warmup):
            out = model.forward_batch(batch, args.loop)
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = model.forward_batch(batch, args.loop)
        sync()
        return time.perf_counter() - t0, out

    def over_ranks(elapsed, G):
        """(every rank's seconds, the job's graph count): one all_gather of two numbers per rank."""
        if not use_dist:
            return [elapsed], G
        mine = torch.tensor([elapsed, float(G)], dtype=torch.float64, device=cdev)
        allr = torch.empty(2 * world, dtype=torch.float64, device=cdev)
        dist.all_gather_into_tensor(allr, mine)
        allr = allr.view(world, 2).cpu()
        return [float(x) for x in allr[:, 0]], int(round(float(allr[:, 1].sum())))

    if args.strong > 0:
        # strong scaling: the job is the fixed problem set 0 .. N_TOTAL - 1 (problem i = seed 1234 + i whatever the rank count);
        # rank r scores the contiguous block shard_range gives it (equal-size graphs: no weights needed)
        from gnnmp.dist import shard_range
        lo, hi = shard_range(args.strong, rank, world)
        G, seed0 = hi - lo, 1234 + lo
    else:
        G, seed0 = args.graphs, 1234 + rank * args.graphs
    graphs, batch = problem_set(G, seed0, args.unique)

    # `value`: W warm-up steps, K timed steps, per-stage events OFF
    elapsed, scores = timed(batch, args.warmup, args.steps)
    # the same K steps once more with the per-stage HIP events on (12 events per step on the forward's own stream): stage split
    # and the roofline kernel's launch time; its step time is reported next to `ms_per_step`, never as `value`
    model.profile(dev, True)
    elapsed_prof, _ = timed(batch, 1, args.steps)
    prof = model.profile_read(dev)
    model.profile(dev, False)
    for k_ in prof:                       # the warm-up step of the profiled loop is in the sums: scale to the K timed steps
        prof[k_] = (prof[k_][0] * args.steps / (args.steps + 1), prof[k_][1] * args.steps // (args.steps + 1))
    # multi-GPU self-checks (they also run with one rank under GNNMP_BENCH_FORCE_DIST=1): how many ranks the collective
    # library really connected (an all_reduce of ones), every rank's own step time (min / max show imbalance; `value` uses
    # the MAX), the job's total graph count
    ranks_seen = 1
    if use_dist:
        ones = torch.ones(1, dtype=torch.float64, device=cdev)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        ranks_seen = int(round(float(ones.item())))
    rank_s, total_graphs = over_ranks(elapsed, G)
    rank_ms = [x / args.steps * 1e3 for x in rank_s]
    elapsed = max(rank_s)
    elapsed_prof = max(over_ranks(elapsed_prof, G)[0])

    # strong-scaling leg of the same run: a FIXED set of --strong-leg problems (seeds 5000 + i) split over the ranks by
    # shard_range; whole-set rate = N_TOTAL * steps / the slowest rank's time.  One driver sweep over N then gives both curves.
    strong_leg = None
    if args.strong == 0 and args.strong_leg >= world and args.strong_leg > 0:
        from gnnmp.dist import shard_range
        lo, hi = shard_range(args.strong_leg, rank, world)
        _, sbatch = problem_set(hi - lo, 5000 + lo)
        s_el, s_scores = timed(sbatch, 2, args.strong_steps)
        s_rank_s, s_total = over_ranks(s_el, hi - lo)
        s_sum = float(s_scores.double().sum().item())
        if use_dist:
            t_ = torch.tensor([s_sum], dtype=torch.float64, device=cdev)
            dist.all_reduce(t_, op=dist.ReduceOp.SUM)
            s_sum = float(t_.item())
        strong_leg = {'scaling': 'strong', 'problems_total': s_total, 'problems_this_rank0': hi - lo, 'steps': args.strong_steps,
                      'ms_per_step': round(max(s_rank_s) / args.strong_steps * 1e3, 4),
                      'graphs_per_s': round(s_total * args.strong_steps / max(s_rank_s), 2),
                      'rank_ms_per_step': [round(x / args.strong_steps * 1e3, 4) for x in s_rank_s],
                      # sum over ranks of the per-rank score sums: depends on the split only through fp64 summation order
                      'result_checksum': s_sum,
                      'what': 'fixed set of %d problems (seed 5000 + i), contiguous shards by gnnmp.dist.shard_range; '
                              'timed like `value` (barrier + synchronize both sides, max over ranks)' % s_total}
        del sbatch, s_scores
        torch.cuda.empty_cache()

    # PCIe-inclusive variant (reported next to, never instead of, `value`): the reference's own forward
    # timer spans H2D + compute + D2H (eval_gnn.py:193-196); here inputs start in pinned host memory and
    # the per-edge scores end in pinned host memory every step.
    e2e = e2e_serial = None
    if args.pcie_steps > 0:
        from gnnmp.serve import BatchPipeline, pin_batch
        host = pin_batch(batch)
        outs = [torch.empty(batch.total_edges, dtype=torch.float32).pin_memory() for _ in range(2)]
        # (a) one batch at a time: H2D, forward, D2H back to back on one stream
        def step_serial():
            b2 = gnnmp.GraphBatch(*(host[k].to(dev, non_blocking=True) for k in
                                    ('v', 'goal', 'obstacles', 'edge_index', 'node_ptr', 'edge_ptr', 'obs_ptr')),
                                  batch.max_obstacles)
            outs[0].copy_(model.forward_batch(b2, args.loop), non_blocking=True)
        for _ in range(3):                               # first touches of the pinned buffers / allocator growth
            step_serial()
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for _ in range(args.pcie_steps):
            step_serial()
        torch.cuda.synchronize(dev)
        e2e_serial = G * args.pcie_steps / (time.perf_counter() - t1)
        # (b) two batches in flight: copy-in / compute / copy-out on three HIP streams (gnnmp.serve.BatchPipeline)
        pipe = BatchPipeline(model, args.loop, host, dev, depth=2)
        for i in range(4):
            pipe.submit(host, outs[i % 2])
        pipe.drain()
        t1 = time.perf_counter()
        for i in range(2 * args.pcie_steps):
            pipe.submit(host, outs[i % 2])
        pipe.drain()
        e2e = G * 2 * args.pcie_steps / (time.perf_counter() - t1)
        if not torch.equal(outs[0], scores.cpu()) or not torch.equal(outs[1], scores.cpu()):
            raise SystemExit('pipelined scores differ from the resident-input run')
        del pipe

    # secondary: two resident batches in flight (stream A scores batch 0 while stream B scores batch 1, each with its own workspace and
    # score buffer): the launch tails of one forward are filled by the other's workgroups.  `value` stays the one-stream figure (its
    # ms_per_step is a step's latency and adds up from the stage times); scores are checked against the one-stream run.
    inflight_rate = None
    if args.inflight_steps > 0 and not use_dist:
        depth = max(2, args.inflight_depth)
        sts = [torch.cuda.Stream(dev) for _ in range(depth)]
        bufs = [(torch.empty(model.workspace_bytes(batch), dtype=torch.uint8, device=dev),
                 torch.empty(max(batch.total_edges, 1), dtype=torch.float32, device=dev)) for _ in range(depth)]
        def two_in_flight(n):
            cur = torch.cuda.current_stream(dev)
            for st_ in sts:
                st_.wait_stream(cur)
            for i in range(n):
                ws_, out_ = bufs[i % depth]
                with torch.cuda.stream(sts[i % depth]):
                    model.forward_batch(batch, args.loop, ws=ws_, out=out_)
            for st_ in sts:
                cur.wait_stream(st_)
        two_in_flight(2 * depth)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        two_in_flight(args.inflight_steps)
        torch.cuda.synchronize(dev)
        inflight_rate = G * args.inflight_steps / (time.perf_counter() - t1)
        for _, out_ in bufs:
            if not torch.equal(out_[:batch.total_edges], scores):
                raise SystemExit('scores of the two-batches-in-flight loop differ from the one-stream run')
        del bufs
        torch.cuda.empty_cache()

    # secondary: drop-in output format (the reference's zero-filled dense [N, N] block per graph, model.py:148-149)
    dense_rate = None
    if args.dense_steps > 0:
        torch.cuda.empty_cache()          # the 1 GB output block must come out of the allocator's cache on every step, not from a fresh device malloc
        model.forward_batch(batch, args.loop, dense=True)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for _ in range(args.dense_steps):
            model.forward_batch(batch, args.loop, dense=True)
        torch.cuda.synchronize(dev)
        dense_rate = G * args.dense_steps / (time.perf_counter() - t1)

    # secondary: the same workload in the opt-in bf16x3 mode (held to the same parity bar, tests/test_explorer_bf16x3.py)
    x3_rate = None
    if args.bf16x3_steps > 0 and args.mlp_dtype == 'fp32':
        model.mlp_dtype = 'bf16x3'
        for _ in range(2):
            model.forward_batch(batch, args.loop)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for _ in range(args.bf16x3_steps):
            model.forward_batch(batch, args.loop)
        torch.cuda.synchronize(dev)
        x3_rate = G * args.bf16x3_steps / (time.perf_counter() - t1)
        model.mlp_dtype = args.mlp_dtype

    # secondary: the call the reference itself makes (eval_gnn.py:194: ONE graph per forward).  Median of 5 blocks of 50
    # back-to-back calls: prebuilt one-graph batch with per-edge scores, and the drop-in module call with the dense
    # [N, N] result.
    single_us = None
    if args.single_steps > 0:
        g0 = graphs[0]
        b1 = model._single(g0['goal'], g0['v'], g0['obstacles'], g0['edge_index'])

        def med(fn):
            for _ in range(10):
                fn()
            ts = []
            for _ in range(5):
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                for _ in range(args.single_steps):
                    fn()
                torch.cuda.synchronize(dev)
                ts.append((time.perf_counter() - t1) / args.single_steps)
            return round(sorted(ts)[2] * 1e6, 1)
        single_us = {'sparse_scores': med(lambda: model.forward_batch(b1, args.loop)),
                     'dense_drop_in_call': med(lambda: model(goal=g0['goal'], loop=args.loop, v=g0['v'], obstacles=g0['obstacles'],
                                                             edge_index=g0['edge_index']))}

    # final result gather (the only collective of the job): per-rank edge scores -> every rank
    # It is timed on its own (barrier + sync on both sides, median of --gather-reps, max over ranks) and NEVER enters `value`:
    # no rank needs another rank's scores to make progress.
    checksum = float(scores.double().sum().item())
    gather_ms = None
    if use_dist:
        from gnnmp.dist import gather_variable
        gsrc = scores if backend == 'nccl' else scores.cpu()   # gloo gathers host tensors
        parts = gather_variable(gsrc)                      # one padded buffer, all_gather_into_tensor (RCCL); also the warm-up
        checksum = float(torch.stack([p_.double().sum() for p_ in parts]).sum().item())
        ts = []
        for _ in range(max(args.gather_reps, 1)):
            sync()
            t1 = time.perf_counter()
            parts = gather_variable(gsrc)
            torch.cuda.synchronize(dev)
            ts.append(time.perf_counter() - t1)
        tg = torch.tensor([sorted(ts)[len(ts) // 2]], dtype=torch.float64, device=cdev)
        dist.all_reduce(tg, op=dist.ReduceOp.MAX)
        gather_ms = round(float(tg.item()) * 1e3, 4)
        del parts

    if rank == 0:
        Ns = [int(g['v'].shape[0]) for g in graphs]
        Es = [int(g['edge_index'].shape[1]) for g in graphs]
        Os = [g['obstacles'].reshape(-1, e['S']).shape[0] for g in graphs]
        flops_batch = sum(algorithmic_flops(n, m, o, e['C'], e['d'], e['S'], args.loop) for n, m, o in zip(Ns, Es, Os))
        bytes_batch = sum(algorithmic_bytes(n, m, o, e['C'], e['S']) for n, m, o in zip(Ns, Es, Os))
        roof, stage_roof = roofline_blocks(args, e, G, prof, Ns, Es, Os)
        is_bf16 = args.mlp_dtype == 'bf16'
        mfma_peak = PEAK_BF16_TFLOPS if is_bf16 else PEAK_FP32_TFLOPS
        shape = (args.env, args.nodes, args.k1, args.mlp_dtype)
        cfg_name = 'BASELINE configs[1]' if shape == ('maze2', 1000, 8, 'fp32') and G == 256 \
            else ('BASELINE configs[2] shape' if shape == ('kuka7', 2000, 10, 'bf16')
                  else ('BASELINE configs[4] shape (explorer half)' if shape == ('kuka14', 5000, 16, 'bf16')
                        else ('BASELINE configs[0] shape, batched' if shape == ('maze2', 200, 6, 'fp32') else 'custom workload')))
        ms_step = elapsed / args.steps * 1e3
        value = total_graphs * args.steps / elapsed          # whole job: the graphs ALL ranks scored per step / the slowest rank's time
        stages = {k: round(v[0] / max(args.steps, 1), 4) for k, v in prof.items()}
        res = {
            'metric': 'RGG graphs/sec (GNN explorer forward), %d-node k=%d' % (args.nodes, args.k1),
            'value': round(value, 2), 'unit': 'graphs/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms_step, 4), 'higher_is_better': True, 'scaling': 'strong' if args.strong > 0 else 'weak',
            'vs_baseline': None,
            'dtype': {'fp32': 'f32', 'bf16': 'bf16', 'bf16x3': 'f32 (3 x bf16 split MFMA operands, fp32 accumulate)'}[args.mlp_dtype],
            'data': 'synthetic',
            'config': {'workload': '%s: %s, batch of %d problems per GPU, %d-node k1=%d RGGs '
                                   '(mean E=%.0f, O=%d), loop=%d, use_obstacles, real %s checkpoint, %s, sparse '
                                   'per-edge scores' % (cfg_name, args.env, G, args.nodes, args.k1, sum(Es) / len(Es), Os[0],
                                                        args.loop, e['ckpt'], args.mlp_dtype),
                       'graphs_per_gpu': G, 'graphs_total': total_graphs,
                       'host': {'cpus_visible': os.cpu_count(), 'cpu_quota': cpu_quota(), 'torch_threads_default': machine_threads,
                                'torch_threads_used': host_threads},
                       'parallelism': 'problem-sharded x%d (%s)' % (world, 'fixed set of %d problems split by shard_range' % args.strong
                                                                  if args.strong > 0 else '%d problems per GPU' % G),
                       # multi-GPU self-checks: ranks the collective library connected, every rank's own ms per step, the result
                       # gather timed on its own (not part of `value`)
                       'ranks_seen': ranks_seen,
                       'rank_ms_per_step': {'min': round(min(rank_ms), 4), 'max': round(max(rank_ms), 4),
                                            'all': [round(x, 4) for x in rank_ms]},
                       'collective_backend': (backend if use_dist else None),
                       'ms_per_step_with_stage_events': round(elapsed_prof / args.steps * 1e3, 4),
                       'timing': '`value` / `ms_per_step`: %d steps with the per-stage HIP events OFF; stage_ms_per_step, stage_roofline and '
                                 'roofline.launch_ms: a second loop of the same %d steps with the events on' % (args.steps, args.steps),
                       'strong_leg': strong_leg,
                       'gather_ms': gather_ms,
                       'gather': None if gather_ms is None else 'per-edge scores of every rank -> every rank: two all_gather_into_tensor '
                                 'calls (lengths, one padded payload of %d floats per rank), median of %d, max over ranks; outside the '
                                 'timed region' % (int(scores.numel()), max(args.gather_reps, 1)),
                       # rank 0's share of the job against the peaks of ONE GPU (rank 0's FLOPs / bytes over the job's step time)
                       'whole_forward': whole_forward_block('%s N=%d k1=%d graphs=%d %s' % (args.env, args.nodes, args.k1, G, args.mlp_dtype),
                                                            flops_batch, bytes_batch, elapsed / args.steps, is_bf16),
                       'stage_ms_per_step': stages, 'stage_roofline': stage_roof, 'result_checksum': checksum,
                       'pcie_inclusive_graphs_per_s_per_gpu': None if e2e is None else round(e2e, 1),
                       'pcie_inclusive_one_batch_at_a_time': None if e2e_serial is None else round(e2e_serial, 1),
                       'dense_output_graphs_per_s_per_gpu': None if dense_rate is None else round(dense_rate, 1),
                       'batches_in_flight_graphs_per_s_per_gpu': None if inflight_rate is None else {'depth': max(2, args.inflight_depth), 'graphs_per_s': round(inflight_rate, 1)},
                       'bf16x3_mode_graphs_per_s_per_gpu': None if x3_rate is None else round(x3_rate, 1),
                       'single_graph_us': single_us},
            'roofline': roof,
        }
        if world == 1 and args.other_configs_steps > 0 and args.strong == 0 and (args.env, args.nodes, args.k1, args.mlp_dtype, args.graphs) == ('maze2', 1000, 8, 'fp32', 256):
            del batch
            torch.cuda.empty_cache()
            res['config']['other_configs_gpu'] = {key: other_config_leg(key, oe, on, ok, og, od, args.loop, args.other_configs_steps, 10, dev)
                                                  for key, oe, on, ok, og, od in OTHER_CONFIGS}
        if world == 1 and args.planner_problems > 0 and (args.env, args.mlp_dtype) == ('maze2', 'fp32'):
            res['config']['planner'] = planner_leg(args.planner_problems, 1024, dev)
        if world == 1 and not args.no_cpu_baseline:
            res['cpu_baseline'] = cpu_baseline(args.env, args.nodes, args.k1, args.cpu_seconds, 1234)
        print(json.dumps(res), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

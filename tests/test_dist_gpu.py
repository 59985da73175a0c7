"""Multi-process paths on the GPU box (one MI355X): (1) bench.py's RCCL code path -- init_process_group('nccl'),
barrier, all_reduce, all_gather -- under torch.distributed.run with one rank, against the plain single-process line;
(2) two host processes sharing the GPU, each evaluating ITS shard of planning problems with the device planner
(planner.eval_gnn_device(shard=(rank, 2))) and gathering the per-problem rows (eval_gnn.py:120-122) with
gnnmp.dist.gather_problem_results over gloo (two ranks cannot share one GPU under RCCL) -- against the
single-process rows and the committed fixture."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, REPO

pytestmark = pytest.mark.gpu
BENCH_ARGS = ['--steps', '3', '--warmup', '1', '--graphs', '32', '--no-cpu-baseline', '--planner-problems', '0', '--pcie-steps', '0', '--dense-steps', '0',
              '--bf16x3-steps', '0', '--single-steps', '0', '--other-configs-steps', '0']


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _json_line(out):
    lines = [ln for ln in out.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_bench_rccl_path_one_rank_equals_plain_run():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    plain = subprocess.run([sys.executable, 'bench.py', '--gpus', '1'] + BENCH_ARGS, cwd=REPO, env=env, capture_output=True,
                           text=True, timeout=600)
    assert plain.returncode == 0, plain.stderr[-2000:]
    a = _json_line(plain.stdout)
    env2 = dict(env, GNNMP_BENCH_FORCE_DIST='1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), 'bench.py', '--gpus', '1'] + BENCH_ARGS
    dist_run = subprocess.run(cmd, cwd=REPO, env=env2, capture_output=True, text=True, timeout=600)
    assert dist_run.returncode == 0, dist_run.stderr[-2000:]
    b = _json_line(dist_run.stdout)
    assert b['n_gpus'] == 1 and a['n_gpus'] == 1
    # the gathered (all_gather over RCCL) scores are the same bytes as the local ones
    assert a['config']['result_checksum'] == b['config']['result_checksum']
    assert b['value'] > 0.5 * a['value']
    # the multi-GPU self-checks of the line (the driver's 8-GPU run reads these): ranks the collective library connected, every
    # rank's own step time, the result gather timed on its own and outside `value`
    ca, cb = a['config'], b['config']
    assert ca['ranks_seen'] == 1 and ca['gather_ms'] is None                    # plain run: no process group, nothing gathered
    assert cb['ranks_seen'] == 1 and cb['gather_ms'] is not None and 0 < cb['gather_ms'] < 50
    assert cb['rank_ms_per_step']['min'] == cb['rank_ms_per_step']['max'] == cb['rank_ms_per_step']['all'][0]
    assert abs(cb['rank_ms_per_step']['max'] - b['ms_per_step']) < 1e-3
    assert cb['graphs_total'] == cb['graphs_per_gpu'] == 32 and b['scaling'] == 'weak'


def test_bench_strong_mode_is_the_same_fixed_problem_set():
    """--strong N_TOTAL: the job is problems 0 .. N_TOTAL - 1 (seed 1234 + i) whatever the rank count; with one rank it is
    exactly the weak run of N_TOTAL graphs (same checksum), labelled `strong`; under the forced RCCL path the gathered
    checksum is the same again.  (The 2- and 3-rank split of the set is covered on CPU: tests/test_dist_gloo.py.)"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    weak = subprocess.run([sys.executable, 'bench.py', '--gpus', '1'] + BENCH_ARGS, cwd=REPO, env=env, capture_output=True, text=True, timeout=600)
    assert weak.returncode == 0, weak.stderr[-2000:]
    args = [x for x in BENCH_ARGS]
    gi = args.index('--graphs')
    del args[gi:gi + 2]
    strong = subprocess.run([sys.executable, 'bench.py', '--gpus', '1', '--strong', '32'] + args, cwd=REPO, env=env, capture_output=True,
                            text=True, timeout=600)
    assert strong.returncode == 0, strong.stderr[-2000:]
    a, b = _json_line(weak.stdout), _json_line(strong.stdout)
    assert b['scaling'] == 'strong' and a['scaling'] == 'weak'
    assert b['config']['graphs_total'] == 32 and b['config']['result_checksum'] == a['config']['result_checksum']
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), 'bench.py', '--gpus', '1', '--strong', '32'] + args
    forced = subprocess.run(cmd, cwd=REPO, env=dict(env, GNNMP_BENCH_FORCE_DIST='1'), capture_output=True, text=True, timeout=600)
    assert forced.returncode == 0, forced.stderr[-2000:]
    c = _json_line(forced.stdout)
    assert c['scaling'] == 'strong' and c['config']['result_checksum'] == a['config']['result_checksum'] and c['config']['gather_ms'] > 0


def test_bench_refuses_world_size_mismatch():
    r = subprocess.run([sys.executable, 'bench.py', '--gpus', '2', '--steps', '1'], cwd=REPO, capture_output=True, text=True,
                       timeout=300, env=dict(os.environ, RANK='0', LOCAL_RANK='0', WORLD_SIZE='1'))
    assert r.returncode != 0 and 'WORLD_SIZE' in (r.stderr + r.stdout)


def test_bench_gpus_2_launches_itself_two_ranks_on_this_gpu():
    """`python bench.py --gpus 2` with no launcher environment (the driver's N = 1 command shape with a larger N): bench.py starts
    its own two ranks.  One GPU here, and RCCL refuses two ranks on one device, so the collectives run over gloo
    (GNNMP_BENCH_BACKEND=gloo, host tensors) while both ranks run the HIP forward on the one GPU.  Weak scaling: rank r owns
    seeds 1234 + 32 r + i, so the gathered checksum is the checksum of a single-rank run over 64 graphs (up to the order of the
    fp64 sum); the strong leg's fixed set gives the same checksum for one and two ranks."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY='0', GNNMP_BENCH_BACKEND='gloo')
    extra = ['--strong-leg', '48', '--strong-steps', '2']
    two = subprocess.run([sys.executable, 'bench.py', '--gpus', '2'] + BENCH_ARGS + extra, cwd=REPO, env=env, capture_output=True,
                         text=True, timeout=900)
    assert two.returncode == 0, two.stderr[-3000:]
    b = _json_line(two.stdout)
    args = [x for x in BENCH_ARGS]
    args[args.index('--graphs') + 1] = '64'
    one = subprocess.run([sys.executable, 'bench.py', '--gpus', '1'] + args + extra, cwd=REPO, env=env, capture_output=True, text=True,
                         timeout=900)
    assert one.returncode == 0, one.stderr[-3000:]
    a = _json_line(one.stdout)
    cb, ca = b['config'], a['config']
    assert b['n_gpus'] == 2 and cb['ranks_seen'] == 2 and cb['collective_backend'] == 'gloo'
    assert cb['graphs_total'] == 64 and cb['graphs_per_gpu'] == 32 and len(cb['rank_ms_per_step']['all']) == 2
    assert abs(cb['result_checksum'] - ca['result_checksum']) <= 1e-9 * abs(ca['result_checksum'])
    assert abs(b['value'] - 64 / (b['ms_per_step'] * 1e-3)) <= 1e-3 * b['value']          # whole job over the slowest rank's time
    sa, sb = ca['strong_leg'], cb['strong_leg']
    assert sa['problems_total'] == sb['problems_total'] == 48 and sb['scaling'] == 'strong' and len(sb['rank_ms_per_step']) == 2
    assert abs(sa['result_checksum'] - sb['result_checksum']) <= 1e-9 * abs(sa['result_checksum'])
    assert cb['gather_ms'] is not None and cb['gather_ms'] > 0


_WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %(repo)r)
import gnnmp
from gnnmp import planner
from gnnmp.dist import gather_problem_results
from gnnmp.maze2d import Maze2D
from gnnmp.weights import load_weights
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo')
with np.load(%(fixture)r) as f:
    r = {k: f[k] for k in f.files}
env = Maze2D(r['maps'], r['init_states'], r['goal_states'])
m = gnnmp.EncoderProcessDecoder(2, 2, 32, 2).eval(); m.load_state_dict(load_weights('weights_maze'))
ms = gnnmp.ModelSmoother(workspace_size=2, config_size=2, embed_size=128, obs_size=6).eval(); ms.load_state_dict(load_weights('smooth_2d_attv3'))
rows = []
planner.eval_gnn_device(env, range(64), m, ms, seed=int(r['seed']), batch=int(r['batch']), k=int(r['k']), device='cuda:0',
                        rows_out=rows, shard=(rank, world))
allrows = gather_problem_results(torch.tensor(np.array(rows, dtype=np.float64).reshape(-1, 7)))
if rank == 0:
    np.save(%(out)r, allrows.numpy())
dist.barrier()
dist.destroy_process_group()
'''


def test_two_processes_shard_the_device_planner(tmp_path):
    fixture = os.path.join(GOLDEN, 'evalset_mazehard_first1000.npz')
    out = str(tmp_path / 'rows.npy')
    script = tmp_path / 'worker.py'
    script.write_text(_WORKER % dict(repo=REPO, fixture=fixture, out=out))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), str(script)]
    r = subprocess.run(cmd, cwd=REPO, capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0'))
    assert r.returncode == 0, r.stderr[-3000:]
    got = np.load(out)
    with np.load(os.path.join(GOLDEN, 'evalrows_device_first64.npz')) as f:
        want = f['rows']
    assert got.shape == want.shape == (64, 7)
    assert np.array_equal(got, want)          # union over ranks == the recorded single-process device run, row by row


def test_bench_two_ranks_rccl():
    """The driver's multi-GPU launch at N = 2 (one rank per GPU over RCCL): needs two visible GPUs and skips on the one-GPU
    test box.  Weak scaling: both ranks own 32 problems of their own; the line's value is the whole-job rate and the gathered
    checksum is the sum of the two ranks' plain single-process checksums (rank r's problems are seeds 1234 + r * G + i)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip('needs two GPUs (RCCL cannot put two ranks on one device); unmeasured on hardware so far')
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), 'bench.py', '--gpus', '2'] + BENCH_ARGS
    r = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _json_line(r.stdout)
    assert line['n_gpus'] == 2 and line['scaling'] == 'weak' and line['value'] > 0
    assert line['config']['ranks_seen'] == 2 and len(line['config']['rank_ms_per_step']['all']) == 2 and line['config']['gather_ms'] > 0


_GATHER_WORKER = r'''
import os, sys
import torch, torch.distributed as dist
sys.path.insert(0, %(repo)r)
import gnnmp
from gnnmp.dist import gather_variable, run_mixed, shard_range
from gnnmp.synth import ENVS, synth_graph
from gnnmp.weights import load_weights
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo')
dev = 'cuda:0'
order = ['maze2', 'kuka7', 'ur5', 'snake7', 'maze2', 'kuka7', 'maze2', 'ur5', 'snake7']
graphs = [synth_graph(env, 150 + 40 * i, 5, seed=70 + i) for i, env in enumerate(order)]          # ragged: 150 .. 470 nodes
lo, hi = shard_range(len(graphs), rank, world, [g['edge_index'].shape[1] for g in graphs])
models = {}
for env in set(order):
    e = ENVS[env]
    models[env] = gnnmp.EncoderProcessDecoder(e['workspace'], e['C'], e['d'], e['S']).eval()
    models[env].load_state_dict(load_weights(e['ckpt']))
mine = [dict(env=order[i], **{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in graphs[i].items()}) for i in range(lo, hi)]
scores = run_mixed(mine, models, loop=3) if mine else []
flat = torch.cat([s.cpu() for s in scores]) if scores else torch.zeros(0)
parts = gather_variable(flat)                                   # all_gather_into_tensor over gloo, ragged lengths
if rank == 0:
    torch.save(dict(parts=parts, cuts=[shard_range(len(graphs), r, world, [g['edge_index'].shape[1] for g in graphs]) for r in range(world)]), %(out)r)
dist.barrier()
dist.destroy_process_group()
'''


def test_two_ranks_shard_a_ragged_mixed_set_and_gather(tmp_path):
    """Two ranks (gloo, sharing the one GPU) split a ragged mixed-family problem set by edge count, score their shards with
    the HIP kernels and gather the per-edge scores with gather_variable; the gathered buffers equal a single-process run."""
    import torch
    import gnnmp
    from conftest import load_weights
    from gnnmp.dist import run_mixed
    from gnnmp.synth import ENVS, synth_graph
    out = str(tmp_path / 'gathered.pt')
    script = tmp_path / 'worker.py'
    script.write_text(_GATHER_WORKER % dict(repo=REPO, out=out))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), str(script)]
    r = subprocess.run(cmd, cwd=REPO, capture_output=True, text=True, timeout=900, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0'))
    assert r.returncode == 0, r.stderr[-3000:]
    got = torch.load(out)
    order = ['maze2', 'kuka7', 'ur5', 'snake7', 'maze2', 'kuka7', 'maze2', 'ur5', 'snake7']
    graphs = [synth_graph(env, 150 + 40 * i, 5, seed=70 + i) for i, env in enumerate(order)]
    models = {}
    for env in set(order):
        e = ENVS[env]
        models[env] = gnnmp.EncoderProcessDecoder(e['workspace'], e['C'], e['d'], e['S']).eval()
        models[env].load_state_dict(load_weights(e['ckpt']))
    problems = [dict(env=order[i], **{k: (v.to('cuda:0') if torch.is_tensor(v) else v) for k, v in g.items()}) for i, g in enumerate(graphs)]
    whole = [s.cpu() for s in run_mixed(problems, models, loop=3)]
    cuts = got['cuts']
    assert cuts[0][0] == 0 and cuts[0][1] == cuts[1][0] and cuts[1][1] == len(graphs) and 0 < cuts[0][1] < len(graphs)
    for (lo, hi), part in zip(cuts, got['parts']):
        assert torch.equal(part, torch.cat(whole[lo:hi]))

"""bench.py as its own launcher (what is sharded and what is gathered: eval_gnn.py:113-122).
The driver's command at N = 1 is `python bench.py --gpus 1 ...`; the same shape with N > 1 and no launcher environment
must start N ranks, print ONE JSON line from rank 0 and hand a failing rank's status back.  No GPU here: the ranks only
connect (gloo) and leave (`--launch-check`); the full 2-rank run on one GPU is tests/test_dist_gpu.py."""
import json
import os
import subprocess
import sys

from conftest import REPO

_CLEAN = ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'GNNMP_BENCH_BACKEND', 'GNNMP_BENCH_FORCE_DIST')


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in _CLEAN}
    env.update(extra)
    return env


def _json_lines(out):
    return [json.loads(ln) for ln in out.splitlines() if ln.startswith('{')]


def test_gpus_2_without_launcher_starts_two_ranks_and_prints_one_line():
    r = subprocess.run([sys.executable, 'bench.py', '--gpus', '2', '--launch-check'], cwd=REPO, env=_env(GNNMP_BENCH_BACKEND='gloo'),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    assert lines[0]['launch_check'] and lines[0]['n_gpus'] == 2 and lines[0]['ranks_seen'] == 2 and lines[0]['self_launched']
    assert 'torch.distributed.run' in r.stderr            # the command it ran is logged on stderr, never on stdout


def test_launcher_path_still_works_and_a_mismatch_is_refused():
    import socket
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '3', '--master-addr', '127.0.0.1',
           '--master-port', str(port), 'bench.py', '--gpus', '3', '--launch-check']
    r = subprocess.run(cmd, cwd=REPO, env=_env(GNNMP_BENCH_BACKEND='gloo'), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]['ranks_seen'] == 3
    # a launcher environment that disagrees with --gpus is an error on every rank (never a silent single-rank run)
    r = subprocess.run([sys.executable, 'bench.py', '--gpus', '2', '--launch-check'], cwd=REPO,
                       env=_env(RANK='0', LOCAL_RANK='0', WORLD_SIZE='1'), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and 'WORLD_SIZE' in (r.stderr + r.stdout) and not _json_lines(r.stdout)


def test_a_failing_rank_fails_the_self_launched_run():
    # no GPU in this container and no backend override: every rank dies in torch.cuda.set_device / RCCL init; the parent must
    # return non-zero and print no JSON line.  (On a GPU box with >= 2 devices this command is the real thing; skip there.)
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip('two GPUs visible: the command would succeed')
    r = subprocess.run([sys.executable, 'bench.py', '--gpus', '2', '--launch-check'], cwd=REPO, env=_env(), capture_output=True,
                       text=True, timeout=600)
    assert r.returncode != 0 and not _json_lines(r.stdout)
    assert 'exit status' in r.stderr

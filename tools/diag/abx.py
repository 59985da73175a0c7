#!/usr/bin/env python
"""A/B of experiment builds / environment switches on a BASELINE shape, run ON THE GPU BOX.
    python tools/diag/abx.py <cfg: 2|3|5|3f|5f> <variant>[,ENV=val...] ...      ("base" = the in-tree libgnnmp.so)
Prints graphs/s, ms/step, the stage split and the result checksum per variant (two passes each, interleaved)."""
import json, os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CFG = {'2': [], '3': '--env kuka7 --nodes 2000 --k1 10 --graphs 64 --mlp-dtype bf16'.split(),
       '5': '--env kuka14 --nodes 5000 --k1 16 --graphs 32 --mlp-dtype bf16'.split(),
       '3f': '--env kuka7 --nodes 2000 --k1 10 --graphs 64'.split(), '5f': '--env kuka14 --nodes 5000 --k1 16 --graphs 32'.split(),
       '2b': '--mlp-dtype bf16'.split()}
cfg, variants = sys.argv[1], sys.argv[2:]
base = [sys.executable, os.path.join(R, 'bench.py'), '--steps', '20', '--warmup', '5', '--no-cpu-baseline', '--pcie-steps', '0', '--dense-steps', '0',
        '--bf16x3-steps', '0', '--single-steps', '0', '--inflight-steps', '0', '--planner-problems', '0', '--strong-leg', '0'] + CFG[cfg]
for rep in range(2):
    for v in variants:
        parts = v.split(',')
        env = dict(os.environ)
        env.pop('GNNMP_LIB', None)
        if parts[0] != 'base':
            env['GNNMP_LIB'] = os.path.join(R, 'gnn-motion-planning_amd', 'libgnnmp_%s.so' % parts[0])
        for kv in parts[1:]:
            k, val = kv.split('=', 1)
            env[k] = val
        r = subprocess.run(base, env=env, capture_output=True, text=True)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
        if r.returncode != 0 or not lines:
            print('cfg%s %-28s FAILED rc=%d: %s' % (cfg, v, r.returncode, r.stderr[-400:]))
            continue
        d = json.loads(lines[-1])
        print('cfg%s %-28s %9.1f graphs/s  ms/step %.4f  stages %s  checksum %s' % (
            cfg, v, d['value'], d['ms_per_step'], d['config']['stage_ms_per_step'], d['config']['result_checksum']), flush=True)

"""BASELINE configs[4] as one job (tools/cfg5_pipeline.py): explorer forward at the kuka14 5000-node shape in bf16 -> waypoint
paths read off the scores -> five smoother forwards (smoother.py:233-246's loop) in bf16, enqueued back to back on one stream.
The pipelined result equals the same stages run as separate synchronised calls byte for byte, run after run; the smoother
leaves the end points of every path where they were (model_smoother.py:139)."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
import cfg5_pipeline  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('n_problems,nodes,k1', [(4, 1200, 8), (8, 5000, 16)])
def test_pipelined_job_equals_separate_calls(n_problems, nodes, k1):
    job = cfg5_pipeline.Cfg5Job(n_problems, nodes, k1)
    a = job.run().clone()
    b = job.run_separate()
    c = job.run()
    torch.cuda.synchronize()
    assert a.shape == (sum(job.counts), 14)
    assert torch.equal(a, b) and torch.equal(a, c)
    assert torch.isfinite(a).all()
    # end points pass through all five iterations (the smoother rewrites path[1:-1] only)
    p0 = job.paths(job.explore())
    off = 0
    for cnt in job.counts:
        assert torch.equal(a[off], p0[off]) and torch.equal(a[off + cnt - 1], p0[off + cnt - 1])
        off += cnt
    assert not torch.equal(a, p0)                                        # the interior waypoints did move

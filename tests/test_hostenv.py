"""gnnmp.hostenv: the CPU count a container may really use, and the cap on torch's intra-op pool (no GPU needed)."""
import os

import torch


def test_cpu_quota_reads_this_container():
    from gnnmp.hostenv import cpu_quota, limit_host_threads
    q = cpu_quota()
    assert 1 <= q <= (os.cpu_count() or 1)
    n0 = torch.get_num_threads()
    assert limit_host_threads(10 ** 6) == n0                  # never raises the count
    torch.set_num_threads(n0)

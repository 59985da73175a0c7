"""How many host CPUs this process may really use, and keeping torch's intra-op pool inside that.

torch sizes its OpenMP pool from the machine's core count (128 threads on the 256-CPU MI355X hosts) and knows nothing about
a cgroup CPU quota (16 CPUs on those hosts' job containers).  The host loop of the planner runs small parallel torch ops
(``graph_build.knn_indices``: cdist + topk) between GPU calls; after every parallel region the pool's workers spin, the
process burns ~11 cores for a one-core loop, the cgroup's CFS bandwidth runs out and the kernel parks EVERY thread of the
process until the next 100 ms period -- observed as stalls of 10-90 ms (in 10 ms steps) landing on whatever call happens to
be executing, e.g. 5 ms per problem on the drop-in forward span of eval_gnn.py:193-196 whose kernels take 0.24 ms
(profiles/r06_dropin_forward.txt: cpu.stat nr_throttled 0 -> 69 over 60 problems; none with the pool inside the quota).
"""
import os
import warnings


def cpu_quota():
    """CPUs this process may use: the smaller of its affinity mask and its cgroup CPU quota (v2 ``cpu.max``, v1
    ``cpu.cfs_quota_us / cpu.cfs_period_us``), at least 1."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            q, p = f.read().split()[:2]
            if q != 'max':
                quota = float(q) / float(p)
    except (OSError, ValueError):
        try:
            with open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us') as f:
                q = float(f.read())
            with open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as f:
                p = float(f.read())
            if q > 0:
                quota = q / p
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = min(n, max(1, int(quota)))
    return max(1, n)


def limit_host_threads(n=None):
    """Cap torch's intra-op thread pool at ``n`` (default: :func:`cpu_quota`); never raises it.  Returns the count in force."""
    import torch
    cap = cpu_quota() if n is None else max(1, int(n))
    if torch.get_num_threads() > cap:
        torch.set_num_threads(cap)
    return torch.get_num_threads()


_warned = [False]


def warn_if_oversubscribed():
    """One warning per process when torch's pool is larger than the CPU quota (called by the host-loop entry points)."""
    import torch
    if not _warned[0] and torch.get_num_threads() > cpu_quota():
        _warned[0] = True
        warnings.warn('torch uses %d intra-op threads but this process may use %d CPUs (affinity / cgroup quota): spinning pool '
                      'workers get the whole process throttled for tens of milliseconds at a time; call '
                      'gnnmp.hostenv.limit_host_threads() or set OMP_NUM_THREADS' % (torch.get_num_threads(), cpu_quota()), stacklevel=3)

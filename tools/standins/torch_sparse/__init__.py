"""Stand-in for torch_sparse: coalesce = sort columns by (row, col), drop duplicates."""
import torch


def coalesce(index, value, m, n, op='add'):
    assert value is None
    key = index[0].to(torch.int64) * n + index[1].to(torch.int64)
    key = torch.unique(key, sorted=True)
    return torch.stack((key // n, key % n), dim=0), None


class SparseTensor:  # imported by nets.py, never used on the hot path
    pass


def set_diag(*a, **k):
    raise NotImplementedError

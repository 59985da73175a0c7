"""Stand-in for torch_scatter: scatter with the 0-for-empty rule."""
import torch

_RED = {'sum': 'sum', 'add': 'sum', 'max': 'amax', 'min': 'amin', 'mean': 'mean'}


def scatter(src, index, dim=0, out=None, dim_size=None, reduce='sum'):
    if dim < 0:
        dim += src.dim()
    assert dim == 0 or src.dim() > dim
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() else 0
    shape = list(src.shape)
    shape[dim] = dim_size
    res = src.new_zeros(shape)
    if index.numel() == 0:
        return res
    view = [1] * src.dim()
    view[dim] = -1
    idx = index.view(view).expand_as(src)
    # include_self=False: untouched slots keep the initial 0 (torch_scatter semantics)
    return res.scatter_reduce(dim, idx, src, reduce=_RED[reduce], include_self=False)


def scatter_add(src, index, dim=0, out=None, dim_size=None):
    return scatter(src, index, dim, out, dim_size, 'sum')


def scatter_mean(src, index, dim=0, out=None, dim_size=None):
    return scatter(src, index, dim, out, dim_size, 'mean')


def scatter_max(src, index, dim=0, out=None, dim_size=None):
    return scatter(src, index, dim, out, dim_size, 'max'), None

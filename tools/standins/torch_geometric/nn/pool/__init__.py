"""knn / knn_graph restated: float64 distances, topk(largest=False).
GNNMP_STANDIN_KNN=input switches to squared distances accumulated in the INPUT dtype, coordinate by coordinate
(what a float32 kNN library such as torch_cluster's CPU path works in), ties to the lower index."""
import os

import torch


def _knn_input_dtype(x, y, k):
    d = torch.zeros(y.shape[0], x.shape[0], dtype=x.dtype)
    for c in range(x.shape[1]):
        df = x[:, c].view(1, -1) - y[:, c].view(-1, 1)
        d = d + df * df
    return torch.sort(d, dim=1, stable=True).indices[:, :k]


def knn(x, y, k, batch_x=None, batch_y=None, **kw):
    """For each row of y the k nearest rows of x. Returns [2, len(y)*min(k,len(x))];
    row 0 indexes y, row 1 indexes x."""
    k = min(k, x.shape[0])
    if os.environ.get('GNNMP_STANDIN_KNN') == 'input':
        nb = _knn_input_dtype(x, y, k)
    else:
        d = torch.cdist(y.to(torch.float64), x.to(torch.float64))
        nb = d.topk(k, dim=1, largest=False).indices
    q = torch.arange(y.shape[0]).view(-1, 1).expand_as(nb)
    return torch.stack((q.reshape(-1), nb.reshape(-1)), dim=0)


def knn_graph(x, k, batch=None, loop=False, flow='source_to_target', **kw):
    """Row 0 = neighbour (source), row 1 = centre (target)."""
    assert flow == 'source_to_target'
    e = knn(x, x, k if loop else k + 1)
    row, col = e[1], e[0]
    if not loop:
        keep = row != col
        row, col = row[keep], col[keep]
    return torch.stack((row, col), dim=0)

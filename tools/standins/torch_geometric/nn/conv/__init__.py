"""Minimal MessagePassing (flow source_to_target, node_dim=-2): SURVEY App. B."""
import inspect
import torch
from torch_scatter import scatter


class MessagePassing(torch.nn.Module):
    def __init__(self, aggr='add', flow='source_to_target', node_dim=-2, **kwargs):
        super().__init__()
        assert flow == 'source_to_target'
        self.aggr = aggr
        self.node_dim = node_dim
        self._msg_params = [p for p in inspect.signature(self.message).parameters]

    def propagate(self, edge_index, size=None, **kwargs):
        src, dst = edge_index[0], edge_index[1]
        n = None
        feed = {}
        for name in self._msg_params:
            if name.endswith('_i') or name.endswith('_j'):
                base = kwargs[name[:-2]]
                pair = base if isinstance(base, (tuple, list)) else (base, base)
                if name.endswith('_j'):
                    feed[name] = pair[0][src]
                else:
                    feed[name] = pair[1][dst]
                    n = pair[1].shape[0]
            else:
                feed[name] = kwargs.get(name)
        if n is None:
            base = kwargs['x']
            n = (base[1] if isinstance(base, (tuple, list)) else base).shape[0]
        msg = self.message(**feed)
        reduce = {'add': 'sum', 'sum': 'sum', 'max': 'max', 'mean': 'mean'}[self.aggr]
        return scatter(msg, dst, dim=0, dim_size=n, reduce=reduce)

    def message(self, x_j):
        return x_j

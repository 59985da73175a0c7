import torch


def add_self_loops(edge_index, edge_attr=None, fill_value=None, num_nodes=None):
    """Append (i, i) for i in 0..num_nodes-1 after the existing columns; no dedup."""
    if num_nodes is None:
        num_nodes = int(edge_index.max()) + 1
    loop = torch.arange(num_nodes, dtype=edge_index.dtype, device=edge_index.device)
    return torch.cat((edge_index, torch.stack((loop, loop), dim=0)), dim=1), None


def _absent(*a, **k):
    raise NotImplementedError


grid = remove_self_loops = softmax = add_remaining_self_loops = _absent

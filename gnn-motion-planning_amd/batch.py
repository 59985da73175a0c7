"""Block-diagonal batches of independent planning graphs (one RGG per environment instance).

The reference scores one graph per call (eval_gnn.py:113-116,194); the batch is what lets one
launch sequence keep the whole GPU busy.  Graphs stay independent: node ids in ``edge_index``
remain local to their graph, membership comes from the ``*_ptr`` prefix arrays."""
import torch


class GraphBatch:
    """Device-resident concatenation of G graphs: v [sumN, C], goal [G, C], obstacles [sumO, S],
    edge_index [2, sumE] (graph-local ids), node_ptr / edge_ptr / obs_ptr int32 [G+1]."""

    def __init__(self, v, goal, obstacles, edge_index, node_ptr, edge_ptr, obs_ptr, max_obstacles, dense_floats=None):
        # the library reads these through raw pointers as dense row-major arrays (include/gnnmp.h): a strided view (a
        # transposed or column-major tensor) is copied here once instead of being misread
        dense = lambda t: t if t is None or t.is_contiguous() else t.contiguous()      # noqa: E731
        self.v, self.goal, self.obstacles, self.edge_index = dense(v), dense(goal), dense(obstacles), dense(edge_index)
        self.node_ptr, self.edge_ptr, self.obs_ptr = dense(node_ptr), dense(edge_ptr), dense(obs_ptr)
        self.max_obstacles = int(max_obstacles)
        # node_ptr None: ONE graph described by the tensor shapes alone (no prefix arrays on the device, gnnmp.h)
        self.n_graphs = 1 if node_ptr is None else int(node_ptr.numel() - 1)
        # sum_g N_g^2 when the host knows it (sizes the dense output without reading node_ptr back from the device)
        self.dense_floats = int(v.shape[0]) ** 2 if (node_ptr is None and dense_floats is None) else dense_floats

    @property
    def total_nodes(self):
        return int(self.v.shape[0])

    @property
    def total_edges(self):
        return int(self.edge_index.shape[1])

    @property
    def total_obstacles(self):
        return int(self.obstacles.shape[0])

    @staticmethod
    def from_graphs(graphs, obs_size, device):
        """``graphs``: iterable of dicts with v [N,C], goal [C], obstacles [O,S] or [O,2,3],
        edge_index [2,E] (any device)."""
        graphs = list(graphs)
        vs = [g['v'].float() for g in graphs]
        goals = [g['goal'].float().reshape(1, -1) for g in graphs]
        obs = [g['obstacles'].float().reshape(-1, obs_size) for g in graphs]
        eis = [g['edge_index'].long() for g in graphs]

        def prefix(counts):
            p = torch.zeros(len(counts) + 1, dtype=torch.int64)
            p[1:] = torch.tensor(counts, dtype=torch.int64).cumsum(0)
            return p.to(torch.int32)

        b = GraphBatch(
            torch.cat(vs).contiguous().to(device), torch.cat(goals).contiguous().to(device),
            torch.cat(obs).contiguous().to(device), torch.cat(eis, dim=1).contiguous().to(device),
            prefix([x.shape[0] for x in vs]).to(device), prefix([x.shape[1] for x in eis]).to(device),
            prefix([x.shape[0] for x in obs]).to(device), max([x.shape[0] for x in obs] + [0]),
            dense_floats=sum(int(x.shape[0]) ** 2 for x in vs))
        return b

    def split_edges(self, scores):
        """Per-graph views of a [sumE] score vector."""
        if self.edge_ptr is None:
            return [scores]
        ptr = self.edge_ptr.tolist()
        return [scores[ptr[i]:ptr[i + 1]] for i in range(self.n_graphs)]

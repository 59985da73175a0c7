from .pool import knn, knn_graph
from .conv import MessagePassing


def _absent(name):
    def f(*a, **k):
        raise NotImplementedError(name)
    f.__name__ = name
    return f


voxel_grid = _absent('voxel_grid')
radius_graph = _absent('radius_graph')
fps = _absent('fps')
radius = _absent('radius')
global_max_pool = _absent('global_max_pool')
knn_interpolate = _absent('knn_interpolate')


class GraphConv:
    pass


class LEConv:
    pass


class GATConv:
    pass

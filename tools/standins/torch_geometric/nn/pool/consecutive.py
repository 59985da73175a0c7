def consecutive_cluster(*a, **k):
    raise NotImplementedError

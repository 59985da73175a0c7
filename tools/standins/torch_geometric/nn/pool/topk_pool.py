def topk(*a, **k):
    raise NotImplementedError

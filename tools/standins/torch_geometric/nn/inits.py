def glorot(*a, **k):
    raise NotImplementedError


def zeros(*a, **k):
    raise NotImplementedError


def reset(*a, **k):
    raise NotImplementedError

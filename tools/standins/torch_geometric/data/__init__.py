import torch


class Data:
    """Attribute bag with .to(device) and .to_dict()."""

    def __init__(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)

    def to(self, device):
        for k, v in list(self.__dict__.items()):
            if torch.is_tensor(v):
                setattr(self, k, v.to(device))
        return self

    def to_dict(self):
        return dict(self.__dict__)

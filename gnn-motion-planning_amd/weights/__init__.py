"""The reference's shipped checkpoints (MIT-licensed data, `data/weights/*.pt` of the reference repository) converted to
.npz by tools/gen_golden.py: every state_dict tensor under its own name, so
``model.load_state_dict(load_weights('weights_maze'), strict=True)`` is the counterpart of
``model.load_state_dict(torch.load('data/weights/weights_maze.pt'))`` (eval_gnn.py:101,104)."""
import os

import numpy as np
import torch

_DIR = os.path.dirname(os.path.abspath(__file__))


def available():
    return sorted(f[:-4] for f in os.listdir(_DIR) if f.endswith('.npz'))


def load_weights(name):
    """Checkpoint ``name`` (e.g. 'weights_maze', 'smooth_2d_attv3') as a state_dict of CPU tensors."""
    path = os.path.join(_DIR, name + '.npz')
    if not os.path.exists(path):
        raise FileNotFoundError('no checkpoint %r (have: %s)' % (name, ', '.join(available())))
    with np.load(path) as f:
        return {k: torch.from_numpy(f[k]) for k in f.files}

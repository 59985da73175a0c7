"""MI355X-native GNN path-explorer / path-smoother inference (hot path of
rainorangelemon/gnn-motion-planning), behind the reference's ``nn.Module`` call boundary.

Host side is Python (tensor glue only); all arithmetic of the forward passes runs in
hand-written HIP kernels for gfx950 inside ``libgnnmp.so`` (C ABI: ``include/gnnmp.h``).
There is no CPU fallback: constructing a model without the library, or calling it with CPU
tensors, raises.
"""
from . import dist, graph_build, hostenv, serve, synth  # noqa: F401  (host-side helpers, no native code needed)
from .batch import GraphBatch  # noqa: F401
from .explorer import EncoderProcessDecoder  # noqa: F401
from .smoother import ModelSmoother, SmoothBatch  # noqa: F401

__all__ = ['graph_build', 'hostenv', 'synth', 'GraphBatch', 'EncoderProcessDecoder', 'ModelSmoother', 'SmoothBatch']

"""The explorer forward makes no allocation and no synchronisation and enqueues everything on the caller's
stream (include/gnnmp.h), so the whole launch sequence can be captured into a HIP graph and replayed.  Checked
here with torch's graph capture around the C-ABI call: the replay must reproduce the eager result bit for bit,
also after the input buffers were overwritten in place with another problem of the same shape."""
import ctypes

import pytest
import torch

from conftest import load_weights
import gnnmp
from gnnmp import _lib
from gnnmp.synth import synth_graph

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')


def test_forward_is_graph_capturable():
    m = gnnmp.EncoderProcessDecoder(2, 2, 32, 2).eval()
    m.load_state_dict(load_weights('weights_maze'))
    g1 = synth_graph('maze2', 256, 6, seed=1)
    g2 = synth_graph('maze2', 256, 6, seed=2)
    n_e = max(g1['edge_index'].shape[1], g2['edge_index'].shape[1])

    def pad_edges(ei):                                  # same edge count for both problems: repeat the last column
        extra = n_e - ei.shape[1]
        return torch.cat((ei, ei[:, -1:].repeat(1, extra)), dim=1) if extra else ei

    b = gnnmp.GraphBatch.from_graphs([dict(g1, edge_index=pad_edges(g1['edge_index']))], 2, DEV)
    eager = m.forward_batch(b, 5).clone()
    h = m._native(DEV)
    cb = m._cbatch(b)
    ws = m._ws
    out = torch.empty_like(eager)
    stream = torch.cuda.Stream(DEV)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        for _ in range(2):                               # warm-up on the side stream (module loading, attributes)
            _lib.check(_lib.lib().gnnmp_explorer_forward(h, ctypes.byref(cb), 5, 1, out.data_ptr(), None, ws.data_ptr(),
                                                         ws.numel(), stream.cuda_stream), 'forward')
        stream.synchronize()
        graph.capture_begin()
        rc = _lib.lib().gnnmp_explorer_forward(h, ctypes.byref(cb), 5, 1, out.data_ptr(), None, ws.data_ptr(), ws.numel(),
                                               torch.cuda.current_stream().cuda_stream)
        graph.capture_end()
    assert rc == 0
    out.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager)
    # new problem of the same shape written into the SAME buffers, then replay
    b.v.copy_(g2['v'].to(DEV))
    b.goal.copy_(g2['goal'].reshape(1, -1).to(DEV))
    b.obstacles.copy_(g2['obstacles'].reshape(-1, 2).to(DEV))
    b.edge_index.copy_(pad_edges(g2['edge_index']).to(DEV))
    graph.replay()
    torch.cuda.synchronize()
    ref = m.forward_batch(b, 5)
    assert torch.equal(out, ref)


def test_capture_helper():
    m = gnnmp.EncoderProcessDecoder(2, 2, 32, 2).eval()
    m.load_state_dict(load_weights('weights_maze'))
    g = synth_graph('maze2', 200, 6, seed=3)
    b = gnnmp.GraphBatch.from_graphs([g], 2, DEV)
    eager = m.forward_batch(b, 5).clone()
    graph, scores = m.capture(b, 5)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(scores, eager)

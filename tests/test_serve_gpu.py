"""gnnmp.serve.BatchPipeline: host-resident batches scored with copy-in / compute / copy-out overlapped on separate
HIP streams must give exactly the scores of the plain device-resident forward, for a stream of batches of different
sizes (every slot reuses its buffers) and however the submissions interleave."""
import pytest
import torch

from conftest import load_weights
import gnnmp
from gnnmp.serve import BatchPipeline, pin_batch
from gnnmp.synth import synth_graph

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def test_pipeline_equals_direct_forward():
    m = gnnmp.EncoderProcessDecoder(2, 2, 32, 2).eval()
    m.load_state_dict(load_weights('weights_maze'))
    graphs = [synth_graph('maze2', 150 + 10 * i, 5, seed=500 + i) for i in range(12)]
    batches = [gnnmp.GraphBatch.from_graphs(graphs[a:b], 2, DEV) for a, b in ((0, 12), (0, 5), (5, 12), (3, 4), (2, 11))]
    direct = [m.forward_batch(b, 5).cpu() for b in batches]
    hosts = [pin_batch(b) for b in batches]
    pipe = BatchPipeline(m, 5, hosts[0], DEV, depth=2)                   # batch 0 is the largest: the template
    outs = [torch.zeros(batches[0].total_edges).pin_memory() for _ in range(len(batches) * 3)]
    tickets = []
    for rep in range(3):
        for i, h in enumerate(hosts):
            tickets.append((pipe.submit(h, outs[rep * len(batches) + i]), rep * len(batches) + i, i))
    pipe.drain()
    for _, o, i in tickets:
        n = direct[i].numel()
        assert torch.equal(outs[o][:n], direct[i]), (o, i)
    with pytest.raises(ValueError):
        big = pin_batch(gnnmp.GraphBatch.from_graphs(graphs + graphs, 2, DEV))
        pipe.submit(big, torch.zeros(2 * batches[0].total_edges).pin_memory())

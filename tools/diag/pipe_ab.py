import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'tests'))
import torch, gnnmp
from gnnmp.weights import load_weights
from gnnmp.synth import ENVS, synth_batch_gpu
from gnnmp.serve import BatchPipeline, pin_batch
dev = torch.device('cuda:0')
e = ENVS['maze2']
graphs = synth_batch_gpu('maze2', 1000, 8, 256, dev, seed0=1234)
batch = gnnmp.GraphBatch.from_graphs(graphs, e['S'], dev)
model = gnnmp.EncoderProcessDecoder(e['workspace'], e['C'], e['d'], e['S']).eval(); model.load_state_dict(load_weights(e['ckpt']))
host = pin_batch(batch)
outs = [torch.empty(batch.total_edges, dtype=torch.float32).pin_memory() for _ in range(2)]
for rep in range(3):
    for sc in (True, False):
        model.status_checks = sc
        pipe = BatchPipeline(model, 5, host, dev, depth=2)
        for i in range(4): pipe.submit(host, outs[i % 2])
        pipe.drain()
        t1 = time.perf_counter()
        for i in range(20): pipe.submit(host, outs[i % 2])
        pipe.drain()
        dt = time.perf_counter() - t1
        # resident
        for _ in range(3): model.forward_batch(batch, 5)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        for _ in range(20): model.forward_batch(batch, 5)
        torch.cuda.synchronize(); dr = time.perf_counter() - t2
        print('status_checks %-5s: pipeline %.1f graphs/s   resident %.1f graphs/s' % (sc, 256 * 20 / dt, 256 * 20 / dr), flush=True)
        del pipe

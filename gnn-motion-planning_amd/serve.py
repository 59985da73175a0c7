"""Batches that arrive in HOST memory: copies and compute overlapped on separate HIP streams.

The explorer forward starts from device pointers (include/gnnmp.h); a caller whose problems live in host memory
pays an H2D copy of the inputs (the int64 ``edge_index`` alone is 46 MB per 256-graph cfg-2 batch) and a D2H copy of
the scores.  :class:`BatchPipeline` keeps ``depth`` batches in flight: batch i+1's inputs travel over PCIe on the
copy-in stream while batch i runs on the compute stream and batch i-1's scores travel back on the copy-out stream;
every slot owns its device input buffers, workspace and score buffer, so nothing is shared between batches in
flight and nothing is allocated after construction.  The reference has no counterpart (it scores one graph per call
and wall-clocks H2D + compute + D2H together, eval_gnn.py:193-196)."""
import torch

from .batch import GraphBatch

_FIELDS = ('v', 'goal', 'obstacles', 'edge_index', 'node_ptr', 'edge_ptr', 'obs_ptr')


def pin_batch(batch):
    """Pinned host copies of a :class:`GraphBatch`'s tensors (what a host-side producer would hand over)."""
    host = {k: getattr(batch, k).cpu().pin_memory() for k in _FIELDS}
    host['max_obstacles'] = batch.max_obstacles
    return host


class BatchPipeline:
    """``submit(host_batch, out_host)`` -> ticket; ``wait(ticket)`` blocks until ``out_host[:sumE]`` holds the scores.
    ``template``: a host batch (dict of pinned tensors, see :func:`pin_batch`) giving the MAXIMUM sizes of every
    field; later batches may be smaller."""

    def __init__(self, model, loop, template, device, depth=2):
        self.model, self.loop, self.device, self.depth = model, int(loop), torch.device(device), int(depth)
        self.s_in, self.s_run, self.s_out = (torch.cuda.Stream(self.device) for _ in range(3))
        self.slots = []
        big = GraphBatch(*(template[k].to(self.device) for k in _FIELDS), template['max_obstacles'])
        ws_bytes = model.workspace_bytes(big)
        for _ in range(self.depth):
            # flat buffers of the maximum size per field; a smaller batch uses a contiguous prefix view
            slot = {k: torch.empty(template[k].numel(), dtype=template[k].dtype, device=self.device) for k in _FIELDS}
            slot['ws'] = torch.empty(ws_bytes, dtype=torch.uint8, device=self.device)
            slot['scores'] = torch.empty(int(template['edge_index'].shape[1]), dtype=torch.float32, device=self.device)
            slot['ev_in'], slot['ev_run'], slot['ev_out'] = (torch.cuda.Event() for _ in range(3))
            slot['busy'] = False
            self.slots.append(slot)
        torch.cuda.synchronize(self.device)
        self._next = 0

    def submit(self, host, out_host):
        slot = self.slots[self._next]
        self._next = (self._next + 1) % self.depth
        if slot['busy']:
            slot['ev_out'].synchronize()                       # the slot's previous batch has left the device
        slot['busy'] = True
        views = {}
        with torch.cuda.stream(self.s_in):
            for k in _FIELDS:
                src = host[k]
                if src.numel() > slot[k].numel():
                    raise ValueError('batch field %s exceeds the pipeline template (%d > %d elements)'
                                     % (k, src.numel(), slot[k].numel()))
                dst = slot[k][:src.numel()].view(src.shape)
                dst.copy_(src, non_blocking=True)
                views[k] = dst
            slot['ev_in'].record(self.s_in)
        with torch.cuda.stream(self.s_run):
            self.s_run.wait_event(slot['ev_in'])
            b = GraphBatch(*(views[k] for k in _FIELDS), host['max_obstacles'])
            scores = self.model.forward_batch(b, self.loop, ws=slot['ws'], out=slot['scores'])
            slot['ev_run'].record(self.s_run)
        with torch.cuda.stream(self.s_out):
            self.s_out.wait_event(slot['ev_run'])
            out_host[:scores.numel()].copy_(scores, non_blocking=True)
            slot['ev_out'].record(self.s_out)
        return slot

    @staticmethod
    def wait(ticket):
        ticket['ev_out'].synchronize()

    def drain(self):
        for slot in self.slots:
            if slot['busy']:
                slot['ev_out'].synchronize()
                slot['busy'] = False

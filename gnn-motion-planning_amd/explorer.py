"""Drop-in counterpart of the reference's ``EncoderProcessDecoder`` (model.py:48-150).

Same constructor signature, same parameter / buffer names (so ``load_state_dict`` of the shipped
``.pt`` files succeeds with ``strict=True``), same ``forward`` keyword signature and the same dense
``[N, N]`` return value -- but the arithmetic runs in libgnnmp.so's HIP kernels (gfx950).  The
``torch.nn`` layers below are parameter containers only; this class never calls them.
"""
import ctypes
import operator
import os

import torch
from torch import nn
from torch.nn import Linear as Lin, ReLU, Sequential as Seq

from . import _lib
from .batch import GraphBatch


class _Attention(nn.Module):                      # parameter names of model.py:153-162
    def __init__(self, d):
        super().__init__()
        self.key = Lin(d, d, bias=False)
        self.query = Lin(d, d, bias=False)
        self.value = Lin(d, d, bias=False)
        self.layer_norm = nn.LayerNorm(d, eps=1e-6)


class _FeedForward(nn.Module):                    # model.py:184-190
    def __init__(self, d):
        super().__init__()
        self.w_1 = Lin(d, d)
        self.w_2 = Lin(d, d)
        self.layer_norm = nn.LayerNorm(d, eps=1e-6)


class _Block(nn.Module):                          # model.py:204-210
    def __init__(self, d):
        super().__init__()
        self.attention = _Attention(d)
        self.map_feed = _FeedForward(d)
        self.obs_feed = _FeedForward(d)


class _MPNN(nn.Module):                           # model.py:22-28
    def __init__(self, d):
        super().__init__()
        self.lin_0 = Seq(Lin(d * 5, d), ReLU(), Lin(d, d))
        self.lin_1 = Lin(d * 2, d)
        self.bn = nn.BatchNorm1d(d)


_VERSION = operator.attrgetter('_version')
# GNNMP_CHECK_EDGE_INDEX=1: validate edge_index against the node counts before every forward (one device read-back per
# call).  Like the reference's tensor indexing on the GPU, the kernels themselves do not check node ids (gnnmp.h).
_CHECK_IDS = os.environ.get('GNNMP_CHECK_EDGE_INDEX', '0') not in ('', '0')
_GSTAT_STRIDE = 17                     # status words per graph (gnnmp.h: gnnmp_explorer_status_words = 17 x n_graphs)


def _check_edge_ids(batch):
    if batch.total_edges == 0:
        return
    ei = batch.edge_index
    if batch.node_ptr is None:
        n_of_edge = batch.total_nodes
    else:
        counts = (batch.node_ptr[1:] - batch.node_ptr[:-1]).long()
        per_graph = (batch.edge_ptr[1:] - batch.edge_ptr[:-1]).long()
        n_of_edge = torch.repeat_interleave(counts, per_graph).unsqueeze(0)
    if bool(((ei < 0) | (ei >= n_of_edge)).any()):
        raise IndexError('edge_index holds node ids outside [0, N_g) of their graph')


class _TrainScores(torch.autograd.Function):
    """Per-edge scores with gradients for the parameters the reference's policy loss trains (train_explorer.py:156-186;
    node_free_code / edge_free_code are detached at model.py:141,142,146, so the obstacle-attention stack gets none).
    Forward and backward run in libgnnmp.so (gnnmp_explorer_train_forward / _backward)."""

    @staticmethod
    def forward(ctx, model, batch, loop, *params):
        dev = batch.v.device
        h = model._native(dev)
        cb = model._cbatch(batch)
        need = ctypes.c_size_t()
        _lib.check(_lib.lib().gnnmp_explorer_train_workspace_bytes(h, ctypes.byref(cb), int(loop), ctypes.byref(need)),
                   'gnnmp_explorer_train_workspace_bytes')
        ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
        scores = torch.empty(batch.total_edges, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            st = torch.cuda.current_stream().cuda_stream
            _lib.check(_lib.lib().gnnmp_explorer_train_forward(h, ctypes.byref(cb), int(loop), 1 if model.use_obstacles else 0,
                                                               scores.data_ptr(), ws.data_ptr(), ws.numel(), st),
                       'gnnmp_explorer_train_forward')
        ctx.model, ctx.batch, ctx.loop, ctx.ws, ctx.handle = model, batch, int(loop), ws, h
        return scores

    @staticmethod
    def backward(ctx, d_scores):
        model, batch, dev = ctx.model, ctx.batch, ctx.batch.v.device
        cb = model._cbatch(batch)
        n = int(_lib.lib().gnnmp_explorer_grad_floats(ctx.handle))
        grad = torch.empty(n, dtype=torch.float32, device=dev)
        d_scores = d_scores.contiguous().float()
        with torch.cuda.device(dev):
            st = torch.cuda.current_stream().cuda_stream
            _lib.check(_lib.lib().gnnmp_explorer_train_backward(ctx.handle, ctypes.byref(cb), ctx.loop, d_scores.data_ptr(),
                                                                grad.data_ptr(), ctx.ws.data_ptr(), ctx.ws.numel(), st),
                       'gnnmp_explorer_train_backward')
        out, off = [], 0
        for (name, numel), t in zip(model._manifest, model._live_weights()):
            # parameters may live on the host (the reference keeps them wherever .to() put them): gradients follow them
            g = grad[off:off + numel].view_as(t).to(t.device) if name.split('.')[0] in TRAINABLE else None
            out.append(g)
            off += numel
        return (None, None, None) + tuple(out)


# top-level modules that receive a gradient from the policy loss (everything else forward() reads is behind a detach)
TRAINABLE = ('node_code', 'edge_code', 'goal_encoder', 'encoder', 'process', 'decoder', 'policy')


class EncoderProcessDecoder(nn.Module):
    """``EncoderProcessDecoder(workspace_size, config_size, embed_size, obs_size, use_obstacles=True)``
    (model.py:49).  ``use_obstacles`` is read at call time (model.py:125; toggled by eval_gnn.py:88)."""

    def __init__(self, workspace_size, config_size, embed_size, obs_size, use_obstacles=True):
        super().__init__()
        _lib.lib()                                  # fail loudly if the native library is absent
        self.workspace = workspace_size
        self.config_size = config_size
        self.obs_size = obs_size
        self.use_obstacles = use_obstacles
        self.embed_size = embed_size
        C, d, S = config_size, embed_size, obs_size
        mlp = lambda i: Seq(Lin(i, d), ReLU(), Lin(d, d))       # noqa: E731
        # --- tensors forward() reads
        self.node_code = mlp(C * 4)
        self.edge_code = mlp(C * 2)
        self.obs_node_code = mlp(S)
        self.obs_edge_code = mlp(S)
        self.node_free_code = mlp(C)
        self.edge_free_code = mlp(C * 2)
        self.node_attentions = nn.ModuleList([_Block(d) for _ in range(3)])
        self.edge_attentions = nn.ModuleList([_Block(d) for _ in range(3)])
        self.goal_encoder = nn.Parameter(torch.rand(d))
        self.encoder = Lin(d * 4, d)
        self.process = _MPNN(d)
        self.decoder = Lin(d * 2, d)
        self.policy = Seq(Lin(d * 3, d), ReLU(), Lin(d, d), ReLU(), Lin(d, 1, bias=False))
        # --- present in the reference's state_dict, never read by forward(); kept so that
        #     load_state_dict(strict=True) accepts the shipped checkpoints (model.py:64-67,79,83-105)
        self.free_code = mlp(C)
        self.collided_code = mlp(C)
        self.env_code = mlp(d * 3)
        self.node_pos = Lin(C, d)
        self.lstm = nn.LSTMCell(d, d)
        self.ln = nn.LayerNorm(d)
        self.bn_node = nn.BatchNorm1d(d)
        self.bn_edge = nn.BatchNorm1d(d)
        self.bn_hi = nn.BatchNorm1d(d)
        self.ln_node = nn.LayerNorm(d)
        self.ln_edge = nn.LayerNorm(d)
        self.ln_hi = nn.LayerNorm(d)
        self.process_cat = Lin(d * 2, d)
        self.value = Seq(Lin(d, d), ReLU(), Lin(d, d), ReLU(), Lin(d, 1))
        self.node_free = Lin(d, 1)
        self.edge_free = Lin(d, 1)
        # precision of the MFMA operands: 'fp32' (exact, the reference's precision) or 'bf16' (BASELINE configs[2],
        # [4]: bf16 operands, fp32 accumulate); not a constructor argument so the reference signature is kept
        self.mlp_dtype = 'fp32'
        # device-side status of a forward (obstacle count beyond the batch's promise, node ids outside their graph): the forward's own
        # kernels write it into a slot of pinned host memory (gnnmp_explorer_forward_ex; no copy, no extra launch), looked at on a later
        # call / in check_status().  True (default) = every forward, the reference's own one-graph call included (an out-of-range
        # edge_index id there is clamped to node 0 by the kernels and would otherwise yield finite, wrong scores where the reference's
        # indexing raises); False = none (the words then go to the workspace, where the blocking gnnmp_explorer_status finds them)
        self.status_checks = True
        self._handle = None
        self._handle_key = None
        self._ws = None
        self._manifest = None
        self._wt = None
        self.register_load_state_dict_post_hook(lambda m, _k: m._drop_handle())

    _warned_train_dispatch = False          # one warning per process when forward() takes the training path implicitly

    # ------------------------------------------------------------------ native handle
    def _drop_handle(self):
        self._handle = None                # _lib.NativeHandle: destroyed with its last reference
        self._handle_key = None

    def __del__(self):
        try:
            self._drop_handle()
        except Exception:
            pass

    def __getstate__(self):                 # copy.deepcopy / pickle: the native handle and buffers stay with the original
        st = self.__dict__.copy()
        st.update(_handle=None, _handle_key=None, _ws=None, _ws_streams={}, _wt=None, _manifest=None)
        st.pop('_watch', None)              # pending status copies (pinned buffers, events, a lock) belong to the original
        return st

    def _dims(self):
        modes = {'fp32': 0, 'bf16': 1, 'bf16x3': 2}
        if self.mlp_dtype not in modes:
            raise ValueError("mlp_dtype must be 'fp32', 'bf16' or 'bf16x3'")
        return _lib.ExplorerDims(self.config_size, self.embed_size, self.obs_size, modes[self.mlp_dtype])

    def _apply(self, fn, *a, **k):          # .to() / .float() / .cuda(): parameters are replaced
        self._drop_handle()
        return super()._apply(fn, *a, **k)

    def _live_weights(self):
        """The state_dict tensors forward() reads, looked up on the live modules every call (so a replaced
        Parameter object is seen); the (module, slot) paths are resolved once."""
        if self._manifest is None:
            self._manifest = _lib.manifest('explorer', self._dims())
            paths = []
            for n, _ in self._manifest:
                mod, _, leaf = n.rpartition('.')
                m = self.get_submodule(mod) if mod else self
                paths.append((m, leaf))
            self._wt = paths
        out = []
        for m, leaf in self._wt:
            t = m._parameters.get(leaf)
            out.append(t if t is not None else m._buffers[leaf])
        return out

    def refresh_weights(self):
        """Drop the packed device copy of the weights; the next forward re-packs from the current parameters.
        Needed only after writes the cache key cannot see (``p.data.copy_(...)`` and other ``.data`` writes do not
        bump ``_version``); ``load_state_dict``, ``.to()``, in-place autograd-visible ops and replacing a Parameter
        are detected automatically."""
        self._drop_handle()

    @staticmethod
    def _device_index(device):
        dev = torch.device(device)
        return dev.index if dev.index is not None else torch.cuda.current_device()

    def _native(self, device):
        """Opaque library handle holding the packed weights on ``device`` (rebuilt when the
        parameters were reloaded, moved, replaced or modified in place)."""
        wt = self._live_weights()
        idx = self._device_index(device)
        # staleness check on every call, kept cheap (C-level loops; ~30 us for the 142 tensors): same tensor objects, same
        # autograd version counters, same device / operand mode
        key = (idx, self.mlp_dtype, list(map(_VERSION, wt)), wt)
        hk = self._handle_key
        if self._handle is not None and hk[0] == idx and hk[1] == key[1] and hk[2] == key[2] and len(hk[3]) == len(wt) \
                and all(map(operator.is_, hk[3], wt)):
            return self._handle
        self._drop_handle()
        for (n, numel), t in zip(self._manifest, wt):
            if t.numel() != numel:
                raise RuntimeError('parameter %s has %d elements, library expects %d' % (n, t.numel(), numel))
        # one concatenation per device the parameters live on, ONE copy to the host for each (a training loop re-packs after
        # every optimizer step: 142 separate .to('cpu') calls were 142 stream synchronisations when the model is on the GPU)
        devs = {t.device for t in wt}
        if len(devs) == 1:
            blob = torch.cat([t.detach().reshape(-1).to(torch.float32) for t in wt]).to('cpu').contiguous()
        else:
            blob = torch.cat([t.detach().to('cpu', torch.float32).reshape(-1) for t in wt]).contiguous()
        h = ctypes.c_void_p()
        dims = self._dims()
        with torch.cuda.device(idx):            # the library also restores the caller's current device itself
            _lib.check(_lib.lib().gnnmp_explorer_create(ctypes.byref(h), ctypes.byref(dims), blob.data_ptr(),
                                                        blob.numel(), idx), 'gnnmp_explorer_create')
        h = _lib.NativeHandle(h, _lib.lib().gnnmp_explorer_destroy)
        self._handle, self._handle_key = h, key
        return h

    @staticmethod
    def _cbatch(b):
        return _lib.Batch(b.n_graphs, b.total_nodes, b.total_edges, b.total_obstacles, b.max_obstacles,
                          b.v.data_ptr(), b.goal.data_ptr(), b.obstacles.data_ptr() if b.total_obstacles else None,
                          b.edge_index.data_ptr() if b.total_edges else None,
                          None if b.node_ptr is None else b.node_ptr.data_ptr(),
                          None if b.edge_ptr is None else b.edge_ptr.data_ptr(),
                          None if b.obs_ptr is None else b.obs_ptr.data_ptr())

    def _workspace(self, h, cb, device):
        need = ctypes.c_size_t()
        _lib.check(_lib.lib().gnnmp_explorer_workspace_bytes(h, ctypes.byref(cb), ctypes.byref(need)),
                   'gnnmp_explorer_workspace_bytes')
        # one buffer per (device, stream): planner workers run forwards of the same module concurrently on their own
        # streams (planner.eval_gnn_device), and a workspace must never be in use on two streams at once
        key = (str(torch.device(device)), torch.cuda.current_stream(device).cuda_stream)
        cache = self.__dict__.setdefault('_ws_streams', {})
        ws = cache.get(key)
        if ws is None or ws.numel() < need.value:
            # grown with head room: a planner calls with a slightly different edge count every time, and every
            # re-allocation is a device malloc of tens of MB (milliseconds)
            grow = 0 if ws is None else need.value // 3
            cache.pop(key, None)
            ws = None
            if len(cache) >= 8:                                # streams come and go: keep the cache bounded
                cache.clear()
            ws = cache[key] = torch.empty(need.value + grow, dtype=torch.uint8, device=device)
        self._ws = ws
        return ws

    # ------------------------------------------------------------------ batched entry points
    @torch.no_grad()
    def workspace_bytes(self, batch):
        """Bytes of workspace one forward over ``batch`` needs (for callers that keep their own buffers)."""
        dev = batch.v.device
        need = ctypes.c_size_t()
        cb = self._cbatch(batch)
        _lib.check(_lib.lib().gnnmp_explorer_workspace_bytes(self._native(dev), ctypes.byref(cb), ctypes.byref(need)),
                   'gnnmp_explorer_workspace_bytes')
        return int(need.value)

    def forward_batch(self, batch, loop, dense=False, ws=None, out=None):
        """Score every edge of a :class:`GraphBatch`.  Returns ``scores [sumE]`` in the batch's
        column order, or ``(scores, dense_blocks)`` with the concatenated zero-filled
        ``P_g[target, source]`` matrices (model.py:148-149) when ``dense``.
        ``ws`` (uint8, 256-byte aligned, >= ``workspace_bytes``) and ``out`` (float32 [>= sumE]) let a caller that
        runs several batches concurrently on different streams give each its own buffers; everything is enqueued
        on the current stream."""
        dev = batch.v.device
        if dev.type != 'cuda':
            raise RuntimeError('gnnmp runs on the GPU only (got %s tensors); there is no CPU fallback' % dev)
        if int(loop) < 1:
            raise ValueError('loop must be >= 1: the reference binds `decode` only inside the loop '
                             '(model.py:139-145)')
        if _CHECK_IDS:
            _check_edge_ids(batch)
        watch = self._status_watch()
        watch.poll()                                   # device-side status of EARLIER forwards whose copy has arrived (no wait)
        h = self._native(dev)
        cb = self._cbatch(batch)
        if ws is None:
            ws = self._workspace(h, cb, dev)
        scores = torch.empty(batch.total_edges, dtype=torch.float32, device=dev) if out is None else out[:batch.total_edges]
        dn = None
        if dense:
            total = batch.dense_floats
            if total is None:                                  # hand-built batch: read the node counts back
                n = (batch.node_ptr[1:] - batch.node_ptr[:-1]).to(torch.int64)
                total = int((n * n).sum())
            dn = torch.empty(total, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            st = torch.cuda.current_stream().cuda_stream
            slot = watch.acquire(_GSTAT_STRIDE * batch.n_graphs, batch.n_graphs, ('EncoderProcessDecoder forward', batch.n_graphs)) \
                if self.status_checks else None
            rc = _lib.lib().gnnmp_explorer_forward_ex(
                h, ctypes.byref(cb), int(loop), 1 if self.use_obstacles else 0, scores.data_ptr(),
                dn.data_ptr() if dense else None, ws.data_ptr(), ws.numel(), st, None if slot is None else slot.words.data_ptr())
            if slot is not None:
                if rc == 0:
                    watch.commit(slot)
                else:
                    watch.release(slot)
            _lib.check(rc, 'gnnmp_explorer_forward')
        return (scores, dn) if dense else scores

    def _status_watch(self):
        w = self.__dict__.get('_watch')
        if w is None:
            w = self.__dict__['_watch'] = _lib.StatusWatch('explorer')
        return w

    def check_status(self, batch=None, ws=None):
        """Wait for the device-side status of every forward issued so far on this module and raise RuntimeError if one of them saw
        a graph with more obstacles than its batch's ``max_obstacles`` (the reference attends over ALL obstacles, model.py:125-130:
        such a forward's scores are wrong) or a node id outside its graph (the reference's indexing would raise).  ``forward``
        itself never waits: it looks at the status slots that have already arrived when the NEXT forward starts.
        ``status_checks = False`` switches the slots off (callers that validate their batches themselves); the status words of a
        forward then stay in its workspace, and ``check_status(batch)`` reads them through the blocking C-ABI call
        (gnnmp_explorer_status) from ``ws`` -- default: this module's workspace for (``batch``'s device, the CURRENT stream), i.e.
        call it on the stream the forward over ``batch`` ran on."""
        self._status_watch().poll(wait=True)
        if batch is not None and not self.status_checks:
            dev = batch.v.device
            if ws is None:
                ws = self.__dict__.get('_ws_streams', {}).get((str(torch.device(dev)), torch.cuda.current_stream(dev).cuda_stream))
            if ws is None:
                raise RuntimeError('check_status(batch): no forward of this module has run on the current stream of %s' % dev)
            cb = self._cbatch(batch)
            first = ctypes.c_int32(-1)
            with torch.cuda.device(dev):
                rc = _lib.lib().gnnmp_explorer_status(self._native(dev), ctypes.byref(cb), ws.data_ptr(), ws.numel(),
                                                     torch.cuda.current_stream().cuda_stream, ctypes.byref(first))
            if rc != 0:
                raise RuntimeError('EncoderProcessDecoder forward: %s (first offending graph: %d)'
                                   % (_lib.lib().gnnmp_status_string(rc).decode(), first.value))

    def capture(self, batch, loop):
        """Capture one forward over ``batch`` into a HIP graph (the C-ABI forward allocates nothing and never
        synchronises).  Returns ``(graph, scores)``: after writing new problem data of the SAME shape into the
        batch's tensors in place, ``graph.replay()`` refreshes ``scores``.  Removes the per-kernel launch
        overhead of the ~25-launch sequence for latency-sensitive single-problem use."""
        dev = batch.v.device
        h = self._native(dev)
        cb = self._cbatch(batch)
        ws = self._workspace(h, cb, dev)
        scores = torch.empty(batch.total_edges, dtype=torch.float32, device=dev)
        use_obs = 1 if self.use_obstacles else 0
        fwd = _lib.lib().gnnmp_explorer_forward
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            _lib.check(fwd(h, ctypes.byref(cb), int(loop), use_obs, scores.data_ptr(), None, ws.data_ptr(), ws.numel(),
                           side.cuda_stream), 'gnnmp_explorer_forward')
            side.synchronize()
            graph.capture_begin()
            rc = fwd(h, ctypes.byref(cb), int(loop), use_obs, scores.data_ptr(), None, ws.data_ptr(), ws.numel(),
                     torch.cuda.current_stream(dev).cuda_stream)
            graph.capture_end()
        _lib.check(rc, 'gnnmp_explorer_forward (capture)')
        torch.cuda.current_stream(dev).wait_stream(side)
        graph._gnnmp_keepalive = (batch, ws, cb)         # the graph holds raw pointers into these
        return graph, scores

    def profile(self, device, enable=True):
        """Switch per-stage HIP-event timing on/off for this module's handle on ``device``."""
        _lib.check(_lib.lib().gnnmp_explorer_profile(self._native(torch.device(device)), 1 if enable else 0),
                   'gnnmp_explorer_profile')

    def profile_read(self, device):
        """{stage: (milliseconds summed, launches)} since the last read (waits for the events)."""
        ms = (ctypes.c_double * len(_lib.STAGES))()
        cnt = (ctypes.c_int64 * len(_lib.STAGES))()
        _lib.check(_lib.lib().gnnmp_explorer_profile_read(self._native(torch.device(device)), ms, cnt),
                   'gnnmp_explorer_profile_read')
        return {n: (ms[i], int(cnt[i])) for i, n in enumerate(_lib.STAGES)}

    def debug_tap(self, batch, which):
        """Intermediate of the LAST forward_batch on this module (tests only; see gnnmp.h)."""
        dev = batch.v.device
        h = self._native(dev)
        cb = self._cbatch(batch)
        n = batch.n_graphs if which == 3 else batch.total_nodes * self.embed_size
        out = torch.empty(n, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            st = torch.cuda.current_stream().cuda_stream
            _lib.check(_lib.lib().gnnmp_explorer_debug_tap(h, ctypes.byref(cb), which, out.data_ptr(),
                                                           self._ws.data_ptr(), self._ws.numel(), st), 'debug_tap')
        return out if which == 3 else out.view(batch.total_nodes, self.embed_size)

    # ------------------------------------------------------------------ training path
    def train_scores(self, batch, loop):
        """Per-edge scores [sumE] of a :class:`GraphBatch` WITH autograd: ``loss(scores).backward()`` fills ``.grad`` of
        node_code, edge_code, goal_encoder, encoder, process, decoder and policy -- the parameters the reference's policy
        loss reaches (train_explorer.py:156-186); the obstacle-attention stack is behind the reference's ``detach()``
        calls (model.py:141,142,146) and gets none.  fp32 only."""
        if self.mlp_dtype != 'fp32':
            raise RuntimeError('training runs in fp32 (mlp_dtype = %r)' % self.mlp_dtype)
        if batch.v.device.type != 'cuda':
            raise RuntimeError('gnnmp runs on the GPU only (got %s tensors); there is no CPU fallback' % batch.v.device)
        self._native(batch.v.device)                      # builds the manifest / packed + raw weights on the device
        return _TrainScores.apply(self, batch, loop, *self._live_weights())

    def forward_train(self, goal, loop, v, obstacles, free=None, collided=None, edge_index=None, k=10, **kwargs):
        """The reference call with gradients (train_explorer.py:156-160): dense ``policy_output[N, N]``; the scatter
        of the per-edge scores into the matrix (model.py:148-149) is done by torch so that indexing / log_softmax on
        the result back-propagate into :meth:`train_scores`."""
        b = self._single(goal, v, obstacles, edge_index)
        scores = self.train_scores(b, loop)
        P = scores.new_zeros(v.shape[0], v.shape[0])
        return P.index_put((edge_index[1], edge_index[0]), scores)

    def _single(self, goal, v, obstacles, edge_index, prefix_arrays=True):
        """One graph as a :class:`GraphBatch`.  ``prefix_arrays=False``: no node_ptr / edge_ptr / obs_ptr tensors -- the
        library takes the single graph's sizes from the totals (gnnmp.h), which saves the three small host-to-device
        copies per call that building them costs (inference entry points only)."""
        dev = v.device
        S = self.obs_size
        obs = obstacles.reshape(-1, S).float() if (obstacles is not None and self.use_obstacles) else \
            torch.zeros(0, S, device=dev)
        if not prefix_arrays:
            return GraphBatch(v.float().contiguous(), goal.reshape(1, -1).float().contiguous(), obs.contiguous(),
                              edge_index.long().contiguous(), None, None, None, obs.shape[0])
        i32 = lambda *a: torch.tensor(a, dtype=torch.int32, device=dev)      # noqa: E731
        return GraphBatch(v.float().contiguous(), goal.reshape(1, -1).float().contiguous(), obs.contiguous(),
                          edge_index.long().contiguous(), i32(0, v.shape[0]), i32(0, edge_index.shape[1]),
                          i32(0, obs.shape[0]), obs.shape[0], dense_floats=int(v.shape[0]) ** 2)

    @torch.no_grad()
    def edge_scores(self, goal, loop, v, obstacles, edge_index, **_ignored):
        """Sparse form of :meth:`forward`: scores [E] in ``edge_index`` column order."""
        return self.forward_batch(self._single(goal, v, obstacles, edge_index, prefix_arrays=False), loop)

    # ------------------------------------------------------------------ reference signature
    def forward(self, goal, loop, v, obstacles, free=None, collided=None, edge_index=None, k=10, **kwargs):
        """Reference call (eval_gnn.py:194): returns the dense ``policy_output[N, N]`` with
        ``P[target, source] = score`` (model.py:148-149).  ``free``, ``collided``, ``k``, ``labels``
        and any other extra keyword are accepted and ignored exactly like the reference does
        (model.py:115).  Like ``ModelSmoother.forward`` the call dispatches on the module's mode: under ``train()`` with
        autograd enabled -- the reference's training loop calls ``model(...)`` and back-propagates through the result
        (train_explorer.py:156-176) -- it is :meth:`forward_train`; under ``eval()`` or ``torch.no_grad()`` (eval_gnn.py:168)
        the inference kernels run."""
        if self.training and torch.is_grad_enabled():
            if self.mlp_dtype != 'fp32':
                # the inference kernels under no_grad would return a tensor that "does not require grad" and the caller's
                # backward() would fail far from the cause
                raise RuntimeError("EncoderProcessDecoder is in training mode with autograd enabled but mlp_dtype = %r: the "
                                   "training path runs in fp32 only.  Call .eval() (or wrap the call in torch.no_grad()) for "
                                   "inference with %s operands, or set mlp_dtype = 'fp32' to train." % (self.mlp_dtype, self.mlp_dtype))
            if not EncoderProcessDecoder._warned_train_dispatch:
                EncoderProcessDecoder._warned_train_dispatch = True
                import warnings
                warnings.warn('gnnmp.EncoderProcessDecoder: the module is in training mode (the default of a freshly constructed '
                              'torch module) and autograd is enabled, so model(...) runs the differentiable training path '
                              '(train_explorer.py:156-176).  Call .eval() or use torch.no_grad() for the inference kernels.',
                              stacklevel=2)
            return self.forward_train(goal, loop, v, obstacles, free, collided, edge_index, k, **kwargs)
        with torch.no_grad():
            _, dn = self.forward_batch(self._single(goal, v, obstacles, edge_index, prefix_arrays=False), loop, dense=True)
            return dn.view(v.shape[0], v.shape[0])

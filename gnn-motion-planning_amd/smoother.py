"""Drop-in counterpart of the reference's ``ModelSmoother`` (model_smoother.py:46-142).

Same constructor signature, parameter / buffer names (``load_state_dict(strict=True)`` of the
shipped ``smooth_*_attv3.pt`` files works) and ``forward`` keyword signature; the arithmetic runs
in libgnnmp.so's HIP kernels.  The ``torch.nn`` layers are parameter containers only.
"""
import ctypes
import operator

import torch
from torch import nn
from torch.nn import Linear as Lin, ReLU, Sequential as Seq

from . import _lib


class _MPNN(nn.Module):                            # model_smoother.py:22-28
    def __init__(self, d):
        super().__init__()
        self.lin_0 = Seq(Lin(d * 3, d), ReLU(), Lin(d, d))
        self.lin_1 = Seq(Lin(d, d), ReLU(), Lin(d, d))


class SmoothBatch:
    """Device-resident concatenation of B smoothing problems (see gnnmp_smooth_batch in gnnmp.h)."""

    def __init__(self, paths, frees, collideds, edge_indexes, device, prefix_arrays=True):
        def prefix(counts):
            p = torch.zeros(len(counts) + 1, dtype=torch.int64)
            p[1:] = torch.tensor(counts, dtype=torch.int64).cumsum(0)
            return p.to(torch.int32).to(device)
        # rows of C coordinates; a problem may have no free or no collided samples ([0, C] or an empty tensor)
        C = int(paths[0].reshape(paths[0].shape[0], -1).shape[1]) if paths[0].shape[0] > 0 else int(paths[0].shape[-1])
        rows = lambda t: t.float().reshape(t.shape[0], C) if t.numel() > 0 else t.float().new_zeros(0, C)  # noqa: E731
        f32 = lambda ts: torch.cat([rows(t) for t in ts]).contiguous().to(device)  # noqa: E731
        self.n = len(paths)
        if self.n == 1:
            # the reference's call (one problem, smoother.py:243): no concatenations, and the four prefix arrays travel
            # as ONE small host-to-device copy instead of four
            one = lambda t: rows(t).contiguous().to(device)  # noqa: E731
            self.path, self.free, self.collided = one(paths[0]), one(frees[0]), one(collideds[0])
            self.edge_index = edge_indexes[0].long().contiguous().to(device)
            if prefix_arrays:
                ptrs = torch.tensor([0, paths[0].shape[0], 0, frees[0].shape[0], 0, collideds[0].shape[0], 0,
                                     edge_indexes[0].shape[1]], dtype=torch.int32).to(device)
                self.path_ptr, self.free_ptr, self.coll_ptr, self.edge_ptr = ptrs[0:2], ptrs[2:4], ptrs[4:6], ptrs[6:8]
            else:       # the library takes the single problem's sizes from the totals (gnnmp.h): no host-to-device copy
                self.path_ptr = self.free_ptr = self.coll_ptr = self.edge_ptr = None
        else:
            self.path, self.free, self.collided = f32(paths), f32(frees), f32(collideds)
            self.edge_index = torch.cat([e.long() for e in edge_indexes], dim=1).contiguous().to(device)
            self.path_ptr = prefix([t.shape[0] for t in paths])
            self.free_ptr = prefix([t.shape[0] for t in frees])
            self.coll_ptr = prefix([t.shape[0] for t in collideds])
            self.edge_ptr = prefix([e.shape[1] for e in edge_indexes])
        self.max_path = max(t.shape[0] for t in paths)
        self.max_samples = max(f.shape[0] + c.shape[0] for f, c in zip(frees, collideds))
        self.max_edges = max(e.shape[1] for e in edge_indexes)
        self.path_counts = [t.shape[0] for t in paths]
        self.caps_from_host = True        # max_* computed from the same host-side counts as the prefix arrays: cannot be exceeded

    @classmethod
    def from_device(cls, path, free, collided, edge_index, path_counts, free_counts, coll_counts, edge_counts):
        """Batch over tensors that already live on the device (no host copies): ``path`` [sum P, C], ``free``,
        ``collided`` float32, ``edge_index`` [2, sum E] int64 with problem-local ids; ``*_counts`` host lists."""
        dev = path.device

        def prefix(counts):
            p = torch.zeros(len(counts) + 1, dtype=torch.int64)
            p[1:] = torch.tensor(counts, dtype=torch.int64).cumsum(0)
            return p.to(torch.int32).to(dev)
        sb = cls.__new__(cls)
        sb.n = len(path_counts)
        sb.path, sb.free, sb.collided, sb.edge_index = path.contiguous(), free.contiguous(), collided.contiguous(), \
            edge_index.contiguous()
        sb.path_ptr, sb.free_ptr, sb.coll_ptr, sb.edge_ptr = prefix(path_counts), prefix(free_counts), \
            prefix(coll_counts), prefix(edge_counts)
        sb.max_path = max(path_counts)
        sb.max_samples = max(f + c for f, c in zip(free_counts, coll_counts))
        sb.max_edges = max(edge_counts)
        sb.path_counts = list(path_counts)
        sb.caps_from_host = True
        return sb


_VERSION = operator.attrgetter('_version')


def _cbatch(sb):
    return _lib.SmoothBatch(sb.n, sb.path.shape[0], sb.free.shape[0], sb.collided.shape[0], sb.edge_index.shape[1],
                            sb.max_path, sb.max_samples, sb.max_edges, sb.path.data_ptr(),
                            sb.free.data_ptr() if sb.free.numel() else None,
                            sb.collided.data_ptr() if sb.collided.numel() else None,
                            sb.edge_index.data_ptr() if sb.edge_index.numel() else None,
                            *((None,) * 4 if sb.path_ptr is None else
                              (sb.path_ptr.data_ptr(), sb.free_ptr.data_ptr(), sb.coll_ptr.data_ptr(), sb.edge_ptr.data_ptr())))


# parameters the reference's training loss reaches (train_smoother.py:33-61): everything forward() reads
SMOOTHER_TRAINABLE = ('node_code.0.weight', 'node_code.0.bias', 'node_code.1.weight', 'node_code.1.bias', 'node_code.3.weight',
                      'node_code.3.bias', 'process.lin_0.0.weight', 'process.lin_0.0.bias', 'process.lin_0.2.weight',
                      'process.lin_0.2.bias', 'process.lin_1.0.weight', 'process.lin_1.0.bias', 'process.lin_1.2.weight',
                      'process.lin_1.2.bias', 'smooth_node.weight', 'smooth_node.bias')


MAX_SAMPLES = 2048          # free + collided samples per problem (the kNN kernel keeps <= 32 samples per lane)
MAX_CANDIDATES = 7500       # caller edges + 10 kNN edges per waypoint of one problem (sorted in LDS)


def _check_limits(sb):
    """The kernels' per-problem limits (gnnmp.h); the reference's planner stays far inside them (500 + 500 samples,
    paths of a few dozen waypoints), direct callers get the limit by name instead of GNNMP_ERR_DIMS."""
    if sb.max_samples > MAX_SAMPLES:
        raise ValueError('a smoothing problem has %d free + collided samples; the HIP smoother takes at most %d per problem'
                         % (sb.max_samples, MAX_SAMPLES))
    if sb.max_edges + 10 * sb.max_path > MAX_CANDIDATES:
        raise ValueError('a smoothing problem has %d caller edges and %d waypoints: edges + 10 * waypoints = %d exceeds the HIP '
                         "smoother's limit of %d candidate edges per problem"
                         % (sb.max_edges, sb.max_path, sb.max_edges + 10 * sb.max_path, MAX_CANDIDATES))


class _TrainSmooth(torch.autograd.Function):
    """New path [P, C] of ONE smoothing problem with gradients for the smoother's parameters, the way the reference
    trains it (train_smoother.py:33-61, model.train()): BatchNorm with batch statistics, gradients through the loop's
    in-place path updates.  Forward and backward run in libgnnmp.so (gnnmp_smoother_train_forward / _backward)."""

    @staticmethod
    def forward(ctx, model, sb, loop, *params):
        dev = sb.path.device
        _check_limits(sb)
        h = model._native(dev, for_training=True)
        cb = _cbatch(sb)
        need = ctypes.c_size_t()
        _lib.check(_lib.lib().gnnmp_smoother_train_workspace_bytes(h, ctypes.byref(cb), int(loop), ctypes.byref(need)),
                   'gnnmp_smoother_train_workspace_bytes')
        ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
        out = torch.empty_like(sb.path)
        stats = torch.zeros(max(int(loop), 1), 2, model.embed_size, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            st = torch.cuda.current_stream().cuda_stream
            _lib.check(_lib.lib().gnnmp_smoother_train_forward(h, ctypes.byref(cb), int(loop), out.data_ptr(), stats.data_ptr(),
                                                               ws.data_ptr(), ws.numel(), st), 'gnnmp_smoother_train_forward')
        ctx.model, ctx.sb, ctx.loop, ctx.ws, ctx.handle, ctx.names = model, sb, int(loop), ws, h, [n for n, _ in model._manifest]
        ctx.mark_non_differentiable(stats)
        return out, stats

    @staticmethod
    def backward(ctx, d_out, _d_stats):
        model, sb, dev = ctx.model, ctx.sb, ctx.sb.path.device
        cb = _cbatch(sb)
        n = int(_lib.lib().gnnmp_smoother_grad_floats(ctx.handle))
        grad = torch.empty(n, dtype=torch.float32, device=dev)
        d_out = d_out.contiguous().float()
        with torch.cuda.device(dev):
            st = torch.cuda.current_stream().cuda_stream
            _lib.check(_lib.lib().gnnmp_smoother_train_backward(ctx.handle, ctypes.byref(cb), ctx.loop, d_out.data_ptr(),
                                                                grad.data_ptr(), ctx.ws.data_ptr(), ctx.ws.numel(), st),
                       'gnnmp_smoother_train_backward')
        sd = model.state_dict(keep_vars=True)
        out, off = [], 0
        for name, numel in model._manifest:
            t = sd[name]
            out.append(grad[off:off + numel].view_as(t).to(t.device) if name in SMOOTHER_TRAINABLE else None)
            off += numel
        return (None, None, None) + tuple(out)


class ModelSmoother(nn.Module):
    """``ModelSmoother(workspace_size, config_size, obs_size, embed_size, scale=1.)``
    (model_smoother.py:51; ``scale=np.max(env.bound)`` only for ur5, str2name.py:40)."""

    def __init__(self, workspace_size, config_size, obs_size, embed_size, scale=1.):
        super().__init__()
        _lib.lib()
        self.workspace = workspace_size
        self.config_size = config_size
        self.obs_size = obs_size
        self.latent_dim = workspace_size
        self.scale = scale
        self.embed_size = embed_size
        C, d, S = config_size, embed_size, obs_size
        self.bn1 = nn.BatchNorm1d(C)
        self.bn2 = nn.BatchNorm1d(d)
        # node_code.1 IS bn2 (model_smoother.py:63,65): the state_dict carries both key prefixes
        self.node_code = Seq(Lin(C + 3, d), self.bn2, ReLU(), Lin(d, d))
        self.process = _MPNN(d)
        self.smooth_node = Lin(d, C)
        # present in the reference's state_dict, never read by forward() (model_smoother.py:66-69,75-76,90,92)
        self.edge_code = Lin(C * 2, d)
        self.obs_code = Lin(S, d)
        self.obs_node_code = Seq(Lin(S, d), ReLU(), Lin(d, d))
        self.node_free_code = Seq(Lin(C, d), ReLU(), Lin(d, d))
        self.goal_encoder = nn.Parameter(torch.rand(d))
        self.node_pos = Lin(C, d)
        self.encoder = Lin(d * 2, d)
        self.decoder = Lin(d * 2, d)
        self.mlp_dtype = 'fp32'            # or 'bf16': MFMA operands only (see EncoderProcessDecoder.mlp_dtype)
        # device-side status of a forward (a problem beyond the batch's max_path / max_samples / max_edges promises gets no edges):
        # 'auto' has the kernels write it to a pinned host slot (gnnmp_smoother_forward_ex) only for batches whose caps were NOT derived
        # from host-side counts (SmoothBatch's own constructors derive them: nothing to check), 'always' for every forward, 'never' for
        # none (the words stay in the workspace: check_status(sb) reads them with the blocking gnnmp_smoother_status)
        self.status_checks = 'auto'
        self._handle = None
        self._handle_key = None
        self._manifest = None
        self._wt = None
        self._ws = None
        self.register_load_state_dict_post_hook(lambda m, _k: m._drop_handle())

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

    def _apply(self, fn, *a, **k):
        self._drop_handle()
        return super()._apply(fn, *a, **k)

    def _dims(self):
        if self.mlp_dtype not in ('fp32', 'bf16'):
            raise ValueError("mlp_dtype must be 'fp32' or 'bf16'")
        return _lib.SmootherDims(self.config_size, self.embed_size, float(self.scale), 1 if self.mlp_dtype == 'bf16' else 0)

    def refresh_weights(self):
        """Drop the packed device copy of the weights (see EncoderProcessDecoder.refresh_weights)."""
        self._drop_handle()

    def _native(self, device, for_training=False):
        """The native handle for ``device``, rebuilt when a weight changed.  ``for_training``: the BatchNorm running statistics
        are left out of the staleness check -- the training forward normalises with batch statistics and never reads them,
        but updates them after every call, which would otherwise re-pack and re-upload all weights per training step.  The
        stored key stays the full one, so the next inference call (which folds those statistics into node_code.0) rebuilds."""
        if self._manifest is None:
            self._manifest = _lib.manifest('smoother', self._dims())
            self._bn_slots = frozenset(i for i, (n, _) in enumerate(self._manifest)
                                       if n.endswith(('running_mean', 'running_var', 'num_batches_tracked')))
        sd = self.state_dict(keep_vars=True)
        wt = [sd[n] for n, _ in self._manifest]
        dev = torch.device(device)
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        key = (idx, float(self.scale), self.mlp_dtype, list(map(_VERSION, wt)), wt)
        hk = self._handle_key
        if self._handle is not None and hk[:3] == key[:3] and len(hk[4]) == len(wt) and all(map(operator.is_, hk[4], wt)):
            if hk[3] == key[3]:
                return self._handle
            if for_training and all(a == b for i, (a, b) in enumerate(zip(hk[3], key[3])) if i not in self._bn_slots):
                return self._handle
        self._drop_handle()
        if len({t.device for t in wt}) == 1:       # one copy to the host instead of one per tensor (see EncoderProcessDecoder._native)
            blob = torch.cat([t.detach().reshape(-1).to(torch.float32) for t in wt]).to('cpu').contiguous()
        else:
            blob = torch.cat([t.detach().to('cpu', torch.float32).reshape(-1) for t in wt]).contiguous()
        h = ctypes.c_void_p()
        dims = self._dims()
        with torch.cuda.device(idx):
            _lib.check(_lib.lib().gnnmp_smoother_create(ctypes.byref(h), ctypes.byref(dims), blob.data_ptr(), blob.numel(),
                                                        idx), 'gnnmp_smoother_create')
        h = _lib.NativeHandle(h, _lib.lib().gnnmp_smoother_destroy)
        self._handle, self._handle_key = h, key
        return h

    @torch.no_grad()
    def forward_batch(self, sb, loop):
        """New waypoints [sum P, C] for a :class:`SmoothBatch`."""
        dev = sb.path.device
        if dev.type != 'cuda':
            raise RuntimeError('gnnmp runs on the GPU only (got %s tensors); there is no CPU fallback' % dev)
        _check_limits(sb)
        watch = self.__dict__.get('_watch')
        if watch is None:
            watch = self.__dict__['_watch'] = _lib.StatusWatch('smoother')
        watch.poll()
        h = self._native(dev)
        cb = _cbatch(sb)
        need = ctypes.c_size_t()
        _lib.check(_lib.lib().gnnmp_smoother_workspace_bytes(h, ctypes.byref(cb), ctypes.byref(need)),
                   'gnnmp_smoother_workspace_bytes')
        # one buffer per (device, stream), see EncoderProcessDecoder._workspace
        key = (str(dev), torch.cuda.current_stream(dev).cuda_stream)
        cache = self.__dict__.setdefault('_ws_streams', {})
        ws = cache.get(key)
        if ws is None or ws.numel() < need.value:
            grow = 0 if ws is None else need.value // 3
            cache.pop(key, None)
            ws = None
            if len(cache) >= 8:
                cache.clear()
            ws = cache[key] = torch.empty(need.value + grow, dtype=torch.uint8, device=dev)
        self._ws = ws
        out = torch.empty_like(sb.path)
        with torch.cuda.device(dev):
            st = torch.cuda.current_stream().cuda_stream
            slot = watch.acquire(sb.n, sb.n, ('ModelSmoother forward', sb.n)) if self._wants_status(sb) else None
            rc = _lib.lib().gnnmp_smoother_forward_ex(h, ctypes.byref(cb), int(loop), out.data_ptr(), ws.data_ptr(), ws.numel(), st,
                                                      None if slot is None else slot.words.data_ptr())
            if slot is not None:
                if rc == 0:
                    watch.commit(slot)
                else:
                    watch.release(slot)
            _lib.check(rc, 'gnnmp_smoother_forward')
        return out

    def _wants_status(self, sb):
        return self.status_checks in ('always', True) or (self.status_checks == 'auto' and not getattr(sb, 'caps_from_host', False))

    def check_status(self, sb=None, ws=None):
        """Wait for the status slots of the forwards issued so far (see ``status_checks``).  Raises RuntimeError when a problem
        exceeded its batch's max_path / max_samples / max_edges: it was smoothed WITHOUT its kNN / chain edges, i.e. its waypoints
        are wrong.  With ``status_checks = False`` the words stay in the workspace and ``check_status(sb)`` reads them through the
        blocking C-ABI call (gnnmp_smoother_status) from ``ws`` (default: this module's workspace for the CURRENT stream); the same
        for a batch whose caps came from host-side counts under 'auto' (no slot was used for it)."""
        w = self.__dict__.get('_watch')
        if w is not None:
            w.poll(wait=True)
        if sb is not None and not self._wants_status(sb):
            dev = sb.path.device
            if ws is None:
                ws = self.__dict__.get('_ws_streams', {}).get((str(dev), torch.cuda.current_stream(dev).cuda_stream))
            if ws is None:
                raise RuntimeError('check_status(sb): no forward of this module has run on the current stream of %s' % dev)
            cb = _cbatch(sb)
            first = ctypes.c_int32(-1)
            with torch.cuda.device(dev):
                rc = _lib.lib().gnnmp_smoother_status(self._native(dev), ctypes.byref(cb), ws.data_ptr(), ws.numel(),
                                                     torch.cuda.current_stream().cuda_stream, ctypes.byref(first))
            if rc != 0:
                raise RuntimeError('ModelSmoother forward: %s (first offending problem: %d)'
                                   % (_lib.lib().gnnmp_status_string(rc).decode(), first.value))

    def forward_train(self, path, free, collided, obstacles=None, edge_index=None, loop=10, **kwargs):
        """The reference's TRAINING call (train_smoother.py:52 under ``model.train()``): new path [P, C] with a
        ``grad_fn``; BatchNorm (node_code.1) normalises with the statistics of this call's node rows in every loop
        iteration and its running statistics are updated like torch.nn.BatchNorm1d does (momentum 0.1, unbiased
        variance, one update per loop iteration).  fp32 only."""
        if self.mlp_dtype != 'fp32':
            raise RuntimeError('training runs in fp32 (mlp_dtype = %r)' % self.mlp_dtype)
        if path.device.type != 'cuda':
            raise RuntimeError('gnnmp runs on the GPU only (got %s tensors); there is no CPU fallback' % path.device)
        sb = SmoothBatch([path], [free], [collided], [edge_index], path.device)
        self._native(path.device, for_training=True)
        sd = self.state_dict(keep_vars=True)
        out, stats = _TrainSmooth.apply(self, sb, loop, *[sd[n] for n, _ in self._manifest])
        bn = self.node_code[1]
        with torch.no_grad():
            for it in range(int(loop)):
                # torch.nn.BatchNorm1d: the counter goes up first; momentum None means a cumulative moving average
                bn.num_batches_tracked += 1
                m = 1.0 / float(bn.num_batches_tracked) if bn.momentum is None else bn.momentum
                bn.running_mean.mul_(1 - m).add_(stats[it, 0].to(bn.running_mean.device), alpha=m)
                bn.running_var.mul_(1 - m).add_(stats[it, 1].to(bn.running_var.device), alpha=m)
        return out

    def forward(self, path, free, collided, obstacles=None, edge_index=None, loop=10, **kwargs):
        """Reference call (smoother.py:243): returns the new path [P, C]; ``obstacles`` and extra
        keywords are accepted and ignored like the reference does; the caller's ``path`` tensor is
        never written (model_smoother.py:118).  In ``train()`` mode with autograd enabled this is the training call
        (:meth:`forward_train`); otherwise the inference kernels run (eval-mode BatchNorm, no graph)."""
        if self.training and torch.is_grad_enabled():
            return self.forward_train(path, free, collided, obstacles, edge_index, loop)
        with torch.no_grad():
            sb = SmoothBatch([path], [free], [collided], [edge_index], path.device, prefix_arrays=False)
            return self.forward_batch(sb, loop)

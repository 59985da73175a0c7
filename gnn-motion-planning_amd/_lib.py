"""ctypes binding of libgnnmp.so (C ABI: include/gnnmp.h).  There is no fallback: if the
library is missing every model constructor raises."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('GNNMP_LIB') or os.path.join(_HERE, 'libgnnmp.so')      # GNNMP_LIB: an experiment build (tools/diag/build_variant.sh)

ABI_VERSION = 4                        # include/gnnmp.h gnnmp_abi_version(): what this binding was written against

c_float_p = ctypes.POINTER(ctypes.c_float)
c_int32_p = ctypes.POINTER(ctypes.c_int32)
c_int64_p = ctypes.POINTER(ctypes.c_int64)


class ExplorerDims(ctypes.Structure):
    _fields_ = [('config_size', ctypes.c_int32), ('embed_size', ctypes.c_int32), ('obs_size', ctypes.c_int32),
                ('mlp_dtype', ctypes.c_int32)]          # 0 = fp32, 1 = bf16 MFMA operands (gnnmp.h)


class Batch(ctypes.Structure):
    _fields_ = [('n_graphs', ctypes.c_int32), ('total_nodes', ctypes.c_int32), ('total_edges', ctypes.c_int32),
                ('total_obstacles', ctypes.c_int32), ('max_obstacles', ctypes.c_int32),
                ('v', ctypes.c_void_p), ('goal', ctypes.c_void_p), ('obstacles', ctypes.c_void_p),
                ('edge_index', ctypes.c_void_p), ('node_ptr', ctypes.c_void_p), ('edge_ptr', ctypes.c_void_p),
                ('obs_ptr', ctypes.c_void_p)]


class SmootherDims(ctypes.Structure):
    _fields_ = [('config_size', ctypes.c_int32), ('embed_size', ctypes.c_int32), ('scale', ctypes.c_float),
                ('mlp_dtype', ctypes.c_int32)]


class SmoothBatch(ctypes.Structure):
    _fields_ = [('n_problems', ctypes.c_int32), ('total_path', ctypes.c_int32), ('total_free', ctypes.c_int32),
                ('total_collided', ctypes.c_int32), ('total_edges', ctypes.c_int32), ('max_path', ctypes.c_int32),
                ('max_samples', ctypes.c_int32), ('max_edges', ctypes.c_int32),
                ('path', ctypes.c_void_p), ('free_pts', ctypes.c_void_p), ('collided', ctypes.c_void_p),
                ('edge_index', ctypes.c_void_p), ('path_ptr', ctypes.c_void_p), ('free_ptr', ctypes.c_void_p),
                ('coll_ptr', ctypes.c_void_p), ('edge_ptr', ctypes.c_void_p)]


STAGES = ('prep', 'obs', 'node_pre', 'edge_pre', 'mp', 'policy')

class GraphBuildBatch(ctypes.Structure):
    _fields_ = [('n_graphs', ctypes.c_int32), ('total_nodes', ctypes.c_int32), ('k1_max', ctypes.c_int32),
                ('config_size', ctypes.c_int32), ('v', ctypes.c_void_p), ('node_ptr', ctypes.c_void_p),
                ('n_free', ctypes.c_void_p), ('k1', ctypes.c_void_p)]


class MazeBatch(ctypes.Structure):
    _fields_ = [('n_problems', ctypes.c_int32), ('total_nodes', ctypes.c_int32), ('total_edges', ctypes.c_int32),
                ('width', ctypes.c_int32), ('v', ctypes.c_void_p), ('node_ptr', ctypes.c_void_p),
                ('edge_ptr', ctypes.c_void_p), ('n_free', ctypes.c_void_p), ('edge_index', ctypes.c_void_p),
                ('scores', ctypes.c_void_p), ('maps', ctypes.c_void_p), ('goal_states', ctypes.c_void_p)]


class MazeResume(ctypes.Structure):
    _fields_ = [('n_explored', ctypes.c_void_p), ('explored', ctypes.c_void_p), ('prev', ctypes.c_void_p),
                ('n_pairs', ctypes.c_void_p), ('pairs', ctypes.c_void_p), ('pair_ptr', ctypes.c_void_p)]


class MazeSampleBatch(ctypes.Structure):
    _fields_ = [('n_problems', ctypes.c_int32), ('width', ctypes.c_int32), ('n_free', ctypes.c_int32), ('n_attempts', ctypes.c_int64),
                ('attempts', ctypes.c_void_p), ('maps', ctypes.c_void_p), ('init_states', ctypes.c_void_p),
                ('goal_states', ctypes.c_void_p)]


_lib = None


def lib():
    """The loaded library (raises RuntimeError with a build hint if it is not there)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            'libgnnmp.so not found at %s -- build it with `python -c "import __graft_entry__ as g; g.build()"` '
            '(hipcc --offload-arch=gfx950).  There is no CPU / PyTorch fallback for the forward passes.' % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    vp, sz = ctypes.c_void_p, ctypes.c_size_t
    L.gnnmp_status_string.restype = ctypes.c_char_p
    L.gnnmp_status_string.argtypes = [ctypes.c_int]
    L.gnnmp_last_hip_error.restype = ctypes.c_char_p
    # the ABI number is checked before anything else is bound: a stale build (an experiment library left in GNNMP_LIB, an
    # in-tree .so older than this binding) would otherwise change results or crash with no indication
    if not hasattr(L, 'gnnmp_abi_version'):
        raise RuntimeError('%s exports no gnnmp_abi_version: not a libgnnmp build this binding can use' % LIB_PATH)
    L.gnnmp_abi_version.restype = ctypes.c_int
    abi = L.gnnmp_abi_version()
    if abi != ABI_VERSION:
        raise RuntimeError('%s has ABI version %d, this binding expects %d -- rebuild it (`python -c "import __graft_entry__ '
                           'as g; g.build(force=True)"`)%s' % (LIB_PATH, abi, ABI_VERSION,
                                                               ' or unset GNNMP_LIB' if os.environ.get('GNNMP_LIB') else ''))
    if os.environ.get('GNNMP_LIB'):
        import sys
        print('[gnnmp] GNNMP_LIB is set: using the experiment build %s (ABI %d) instead of the in-tree libgnnmp.so'
              % (LIB_PATH, abi), file=sys.stderr, flush=True)
    L.gnnmp_explorer_manifest.argtypes = [ctypes.POINTER(ExplorerDims), ctypes.c_int, ctypes.c_char_p, sz, c_int64_p]
    L.gnnmp_explorer_create.argtypes = [ctypes.POINTER(vp), ctypes.POINTER(ExplorerDims), vp, sz, ctypes.c_int]
    L.gnnmp_explorer_destroy.argtypes = [vp]
    L.gnnmp_explorer_workspace_bytes.argtypes = [vp, ctypes.POINTER(Batch), ctypes.POINTER(sz)]
    L.gnnmp_explorer_forward.argtypes = [vp, ctypes.POINTER(Batch), ctypes.c_int, ctypes.c_int, vp, vp, vp, sz, vp]
    L.gnnmp_explorer_debug_tap.argtypes = [vp, ctypes.POINTER(Batch), ctypes.c_int, vp, vp, sz, vp]
    L.gnnmp_smoother_grad_floats.restype = ctypes.c_int64
    L.gnnmp_smoother_grad_floats.argtypes = [vp]
    L.gnnmp_smoother_train_workspace_bytes.argtypes = [vp, ctypes.POINTER(SmoothBatch), ctypes.c_int, ctypes.POINTER(sz)]
    L.gnnmp_smoother_train_forward.argtypes = [vp, ctypes.POINTER(SmoothBatch), ctypes.c_int, vp, vp, vp, sz, vp]
    L.gnnmp_smoother_train_backward.argtypes = [vp, ctypes.POINTER(SmoothBatch), ctypes.c_int, vp, vp, vp, sz, vp]
    L.gnnmp_explorer_grad_floats.restype = ctypes.c_int64
    L.gnnmp_explorer_grad_floats.argtypes = [vp]
    L.gnnmp_explorer_train_workspace_bytes.argtypes = [vp, ctypes.POINTER(Batch), ctypes.c_int, ctypes.POINTER(sz)]
    L.gnnmp_explorer_train_forward.argtypes = [vp, ctypes.POINTER(Batch), ctypes.c_int, ctypes.c_int, vp, vp, sz, vp]
    L.gnnmp_explorer_train_backward.argtypes = [vp, ctypes.POINTER(Batch), ctypes.c_int, vp, vp, vp, sz, vp]
    L.gnnmp_explorer_forward_ex.argtypes = [vp, ctypes.POINTER(Batch), ctypes.c_int, ctypes.c_int, vp, vp, vp, sz, vp, vp]
    L.gnnmp_explorer_status_words.argtypes = [ctypes.POINTER(Batch), ctypes.POINTER(sz)]
    L.gnnmp_explorer_status.argtypes = [vp, ctypes.POINTER(Batch), vp, sz, vp, c_int32_p]
    L.gnnmp_status_copy.argtypes = [vp, vp, ctypes.c_int32, vp]
    L.gnnmp_explorer_status_region.argtypes = [vp, ctypes.POINTER(Batch), ctypes.POINTER(sz), ctypes.POINTER(sz)]
    L.gnnmp_explorer_status_decode.argtypes = [vp, ctypes.c_int, c_int32_p]
    L.gnnmp_smoother_status.argtypes = [vp, ctypes.POINTER(SmoothBatch), vp, sz, vp, c_int32_p]
    L.gnnmp_smoother_status_region.argtypes = [vp, ctypes.POINTER(SmoothBatch), ctypes.POINTER(sz), ctypes.POINTER(sz)]
    L.gnnmp_smoother_status_decode.argtypes = [vp, ctypes.c_int, c_int32_p]
    L.gnnmp_explorer_profile.argtypes = [vp, ctypes.c_int]
    L.gnnmp_explorer_profile_read.argtypes = [vp, ctypes.POINTER(ctypes.c_double), c_int64_p]
    L.gnnmp_graph_workspace_bytes.argtypes = [ctypes.POINTER(GraphBuildBatch), ctypes.POINTER(sz)]
    L.gnnmp_graph_build.argtypes = [ctypes.POINTER(GraphBuildBatch), vp, ctypes.c_int64, vp, vp, sz, vp]
    L.gnnmp_maze_explore_workspace_bytes.argtypes = [ctypes.POINTER(MazeBatch), ctypes.POINTER(sz)]
    L.gnnmp_maze_explore.argtypes = [ctypes.POINTER(MazeBatch), vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]
    L.gnnmp_maze_explore_ex.argtypes = [ctypes.POINTER(MazeBatch), ctypes.c_int32, ctypes.POINTER(MazeResume), vp, vp, vp, vp, vp,
                                        vp, vp, vp, vp, vp, sz, vp]
    L.gnnmp_maze_sample.argtypes = [ctypes.POINTER(MazeSampleBatch), vp, vp, vp, vp, vp, vp]
    L.gnnmp_maze_steer.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, vp, vp, vp, vp, vp, vp, vp, vp]
    L.gnnmp_pack_a_tiles.restype = ctypes.c_int64
    L.gnnmp_pack_a_tiles.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]
    L.gnnmp_pack_a_small.restype = ctypes.c_int64
    L.gnnmp_pack_a_small.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]
    L.gnnmp_pack_f64_ops.restype = ctypes.c_int64
    L.gnnmp_pack_f64_ops.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]
    L.gnnmp_pack_vec.restype = ctypes.c_int64
    L.gnnmp_pack_vec.argtypes = [vp, ctypes.c_int, vp]
    if hasattr(L, 'gnnmp_smoother_create'):
        L.gnnmp_smoother_manifest.argtypes = [ctypes.POINTER(SmootherDims), ctypes.c_int, ctypes.c_char_p, sz, c_int64_p]
        L.gnnmp_smoother_create.argtypes = [ctypes.POINTER(vp), ctypes.POINTER(SmootherDims), vp, sz, ctypes.c_int]
        L.gnnmp_smoother_destroy.argtypes = [vp]
        L.gnnmp_smoother_workspace_bytes.argtypes = [vp, ctypes.POINTER(SmoothBatch), ctypes.POINTER(sz)]
        L.gnnmp_smoother_forward.argtypes = [vp, ctypes.POINTER(SmoothBatch), ctypes.c_int, vp, vp, sz, vp]
        L.gnnmp_smoother_forward_ex.argtypes = [vp, ctypes.POINTER(SmoothBatch), ctypes.c_int, vp, vp, sz, vp, vp]
    _lib = L
    return L


def check(status, what):
    if status != 0:
        L = lib()
        msg = L.gnnmp_status_string(status).decode()
        hip = L.gnnmp_last_hip_error().decode()
        raise RuntimeError('%s failed: %s%s' % (what, msg, (' [' + hip + ']') if hip else ''))


class StatusWatch:
    """Non-blocking reader of the device-side status of forwards (gnnmp.h: gnnmp_*_forward_ex / _status_decode).

    forward() never synchronises, so what only the device can see -- a graph with more obstacles than the batch promised, a
    smoothing problem beyond its caps, a node id outside its graph -- is written by the forward's own kernels into a slot of
    PINNED host memory handed to gnnmp_*_forward_ex (no copy, no extra launch).  ``acquire`` picks the next slot of a ring that is
    allocated once (a slot's buffer only ever grows), ``commit`` records the slot's event behind the forward, ``poll`` decodes the
    slots whose event has completed and raises RuntimeError naming every bad forward among them -- called at the start of the
    module's NEXT forward (no wait) and by ``check_status()`` (waits).  When all ``depth`` slots are pending the oldest is waited
    for, so an error can lag by at most ``depth`` forwards.  Slots that have not completed stay pending across an error (their
    kernels may still be writing them); nothing is ever handed back to an allocator while a forward could write it."""

    class _Slot:
        __slots__ = ('words', 'event', 'dev', 'n', 'what')

        def __init__(self):
            self.words, self.event, self.dev, self.n, self.what = None, None, -1, 0, ''

    def __init__(self, kind, depth=32):
        import threading
        self.kind, self.depth = kind, depth
        self.free = [StatusWatch._Slot() for _ in range(depth)]
        self.order = []                                 # pending slots, oldest first
        self.lock = threading.Lock()                    # planner workers run forwards of one module from several host threads

    def acquire(self, n_words, n, what):
        """A slot of >= n_words pinned ints for the forward about to be enqueued, or None while the stream is being captured."""
        import torch
        if torch.cuda.is_current_stream_capturing():    # a caller capturing forwards into a graph: no host-side bookkeeping inside
            return None
        while True:
            with self.lock:
                slot = self.free.pop() if self.free else None
            if slot is not None:
                break
            self.poll(wait_oldest=True)
        if slot.words is None or slot.words.numel() < n_words:
            slot.words = torch.empty(max(int(n_words), 64), dtype=torch.int32, pin_memory=True)       # only ever grows
        dev = torch.cuda.current_device()
        if slot.event is None or slot.dev != dev:       # an event belongs to the device it was first recorded on
            slot.event, slot.dev = torch.cuda.Event(), dev
        slot.n, slot.what = n, what
        return slot

    def commit(self, slot):
        slot.event.record()                             # current stream = the one the forward was enqueued on
        with self.lock:
            self.order.append(slot)

    def release(self, slot):                            # the forward call itself failed: nothing was enqueued
        with self.lock:
            self.free.append(slot)

    def poll(self, wait=False, wait_oldest=False):
        import torch
        if torch.cuda.is_current_stream_capturing():    # hipEventQuery is not capture-safe
            return
        bad = []
        while True:
            with self.lock:
                if not self.order:
                    break
                slot = self.order[0]
                if not (wait or wait_oldest) and not slot.event.query():
                    break
                self.order.pop(0)
            if wait or wait_oldest:
                slot.event.synchronize()
            wait_oldest = False
            first = ctypes.c_int32(-1)
            fn = lib().gnnmp_explorer_status_decode if self.kind == 'explorer' else lib().gnnmp_smoother_status_decode
            rc = fn(slot.words.data_ptr(), slot.n, ctypes.byref(first))
            if rc != 0:
                bad.append('%s (%d %ss): %s (first offending %s: %d)' % (slot.what[0], slot.what[1], 'graph' if self.kind == 'explorer' else 'problem', lib().gnnmp_status_string(rc).decode(),
                                                               'graph' if self.kind == 'explorer' else 'problem', first.value))
            with self.lock:
                self.free.append(slot)
        if bad:
            raise RuntimeError('; '.join(bad) + ' -- the results of %s wrong' % ('that forward are' if len(bad) == 1 else 'those forwards are'))


def manifest(kind, dims):
    """[(reference parameter name, numel)] in blob order."""
    L = lib()
    fn = L.gnnmp_explorer_manifest if kind == 'explorer' else L.gnnmp_smoother_manifest
    n = fn(ctypes.byref(dims), -1, None, 0, None)
    if n < 0:
        check(n, 'gnnmp_%s_manifest' % kind)
    out = []
    buf = ctypes.create_string_buffer(256)
    numel = ctypes.c_int64()
    for i in range(n):
        check(fn(ctypes.byref(dims), i, buf, 256, ctypes.byref(numel)), 'gnnmp_%s_manifest' % kind)
        out.append((buf.value.decode(), int(numel.value)))
    return out


class NativeHandle:
    """Owns one gnnmp_*_create handle: destroyed when the last reference goes (the module drops its reference when the
    weights change; an autograd graph keeps the handle its forward ran with until its backward is done)."""

    def __init__(self, ptr, destroy):
        self._as_parameter_ = ptr            # ctypes passes this where a void* is expected
        self._destroy = destroy

    def __del__(self):
        try:
            self._destroy(self._as_parameter_)
        except Exception:
            pass

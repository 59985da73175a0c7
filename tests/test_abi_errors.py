"""Error behaviour of the C ABI that needs no GPU: every entry point validates its arguments before touching
the device and reports through the negative status codes of include/gnnmp.h (no exception, no crash)."""
import ctypes

import pytest

from gnnmp import _lib

OK, ERR_NULL, ERR_DIMS, ERR_WEIGHTS, ERR_WORKSPACE, ERR_HIP, ERR_ARG = 0, -1, -2, -3, -4, -5, -6


@pytest.fixture(scope='module')
def L():
    return _lib.lib()


def test_manifest_rejects_unsupported_dims(L):
    name, numel = ctypes.create_string_buffer(256), ctypes.c_int64()
    good = _lib.ExplorerDims(2, 32, 2, 0)
    n = L.gnnmp_explorer_manifest(ctypes.byref(good), -1, None, 0, None)
    assert n == 142
    assert L.gnnmp_explorer_manifest(ctypes.byref(good), 0, name, 256, ctypes.byref(numel)) >= 0 and numel.value > 0
    for bad in (_lib.ExplorerDims(2, 48, 2, 0), _lib.ExplorerDims(0, 32, 2, 0), _lib.ExplorerDims(2, 32, 2, 7)):
        assert L.gnnmp_explorer_manifest(ctypes.byref(bad), -1, None, 0, None) == ERR_DIMS
    assert L.gnnmp_explorer_manifest(None, -1, None, 0, None) == ERR_NULL
    sm_bad = _lib.SmootherDims(2, 100, 1.0, 0)
    assert L.gnnmp_smoother_manifest(ctypes.byref(sm_bad), -1, None, 0, None) == ERR_DIMS


def test_create_rejects_wrong_blob_before_touching_the_device(L):
    dims = _lib.ExplorerDims(2, 32, 2, 0)
    h = ctypes.c_void_p()
    blob = (ctypes.c_float * 16)()
    assert L.gnnmp_explorer_create(ctypes.byref(h), ctypes.byref(dims), blob, 16, 0) == ERR_WEIGHTS
    assert L.gnnmp_explorer_create(None, ctypes.byref(dims), blob, 16, 0) == ERR_NULL
    assert L.gnnmp_explorer_create(ctypes.byref(h), ctypes.byref(dims), None, 16, 0) == ERR_NULL


def test_null_and_range_checks(L):
    need = ctypes.c_size_t()
    assert L.gnnmp_explorer_workspace_bytes(None, None, ctypes.byref(need)) == ERR_NULL
    gb = _lib.GraphBuildBatch(1, -5, 4, 2, None, None, None, None)
    assert L.gnnmp_graph_workspace_bytes(ctypes.byref(gb), ctypes.byref(need)) == ERR_ARG
    assert L.gnnmp_graph_workspace_bytes(None, ctypes.byref(need)) == ERR_NULL
    mb = _lib.MazeBatch(0, 10, 10, 15, None, None, None, None, None, None, None, None)
    assert L.gnnmp_maze_explore_workspace_bytes(ctypes.byref(mb), ctypes.byref(need)) == ERR_ARG
    assert L.gnnmp_maze_steer(1, 4, 15, None, None, None, None, None, None, None, None) == ERR_NULL
    fake = ctypes.c_void_p(4096)
    assert L.gnnmp_maze_steer(0, 4, 15, fake, fake, fake, fake, fake, fake, fake, None) == ERR_ARG
    # the steered path may not alias its inputs
    assert L.gnnmp_maze_steer(1, 4, 15, fake, fake, fake, fake, fake, fake, fake, None) == ERR_ARG

"""The C-ABI library loads without a GPU and exports every symbol include/gnnmp.h declares."""
import ctypes
import os
import re

import gnnmp  # noqa: F401
from gnnmp import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(REPO, 'include', 'gnnmp.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(gnnmp_[a-z0-9_]+)\s*\(', src)))


def test_every_declared_symbol_is_exported():
    L = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_functions()
    assert len(names) >= 10
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_status_strings_and_manifest():
    L = _lib.lib()
    assert L.gnnmp_abi_version() >= 1
    assert L.gnnmp_status_string(0) == b'ok'
    assert b'dimension' in L.gnnmp_status_string(-2)
    man = _lib.manifest('explorer', _lib.ExplorerDims(2, 32, 2))
    assert len(man) == 142 and sum(n for _, n in man) == 70880
    man = _lib.manifest('explorer', _lib.ExplorerDims(7, 64, 6))
    assert sum(n for _, n in man) == 280320
    # unsupported embed size is refused, not silently accepted
    assert L.gnnmp_explorer_manifest(ctypes.byref(_lib.ExplorerDims(2, 48, 2)), -1, None, 0, None) == -2


def test_ctypes_structs_mirror_the_header():
    """Field names and order of every struct in include/gnnmp.h against the ctypes mirrors in gnnmp/_lib.py
    (a silent drift here corrupts every call)."""
    import ctypes
    import re
    from gnnmp import _lib
    text = open(os.path.join(REPO, 'include', 'gnnmp.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    structs = {}
    for body, name in re.findall(r'typedef struct\s*\{(.*?)\}\s*(\w+);', text, flags=re.S):
        fields = []
        for decl in body.split(';'):
            decl = decl.strip()
            if not decl:
                continue
            # "const int32_t *node_ptr, *edge_ptr, *n_free" -> names after stripping the type and stars
            names = re.findall(r'[\*\s,](\w+)\s*(?=,|$)', ' ' + decl)
            first = decl.split(',')[0].split()[-1].lstrip('*')
            rest = [x.strip().lstrip('*') for x in decl.split(',')[1:]]
            fields.extend([first] + rest)
            assert names
        structs[name] = fields
    mirrors = {'gnnmp_explorer_dims': _lib.ExplorerDims, 'gnnmp_batch': _lib.Batch, 'gnnmp_smoother_dims': _lib.SmootherDims,
               'gnnmp_smooth_batch': _lib.SmoothBatch, 'gnnmp_graph_batch': _lib.GraphBuildBatch,
               'gnnmp_maze_batch': _lib.MazeBatch, 'gnnmp_maze_resume': _lib.MazeResume,
               'gnnmp_maze_sample_batch': _lib.MazeSampleBatch}
    assert set(mirrors) <= set(structs), sorted(structs)
    for cname, cls in mirrors.items():
        assert [f[0] for f in cls._fields_] == structs[cname], cname
    assert ctypes.sizeof(_lib.ExplorerDims) == 16 and ctypes.sizeof(_lib.SmootherDims) == 16

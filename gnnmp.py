"""Import alias: ``import gnnmp`` loads the package that lives in ``gnn-motion-planning_amd/``.

The directory name is fixed by the project layout and is not a valid Python identifier, so
this shim registers it under the importable name ``gnnmp`` (sub-modules resolve through
``submodule_search_locations``, e.g. ``import gnnmp.explorer``).
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'gnn-motion-planning_amd')
_spec = importlib.util.spec_from_file_location(
    'gnnmp', os.path.join(_dir, '__init__.py'), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules['gnnmp'] = _mod
_spec.loader.exec_module(_mod)

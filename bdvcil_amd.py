"""Importable alias for the package directory ``background-debiased-video-cil_amd/`` (a hyphenated
directory name cannot be imported directly).  ``import bdvcil_amd`` loads that directory as the
package ``bdvcil_amd``; sub-modules resolve as ``bdvcil_amd.<name>`` (so pickling under
``ddp_spawn`` works)."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), 'background-debiased-video-cil_amd')
_spec = _ilu.spec_from_file_location(__name__, _os.path.join(_dir, '__init__.py'), submodule_search_locations=[_dir])
_mod = _ilu.module_from_spec(_spec)
_sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)

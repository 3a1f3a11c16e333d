"""Import alias: the package directory is ``mpc-code_amd/`` (not a valid identifier).

``import mpc_code_amd`` from the repo root executes this stub, which loads that directory
as the package ``mpc_code_amd`` and replaces itself in ``sys.modules``.
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mpc-code_amd")
_spec = importlib.util.spec_from_file_location(
    "mpc_code_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_pkg = importlib.util.module_from_spec(_spec)
sys.modules["mpc_code_amd"] = _pkg
_spec.loader.exec_module(_pkg)

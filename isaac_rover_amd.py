"""Import shim: ``import isaac_rover_amd`` loads the package directory ``isaac_rover_2.0_amd/``.

The package directory carries the reference's name (a dot is not importable), so this module
replaces itself in ``sys.modules`` with the real package loaded from that directory.
"""
import importlib.util
import os
import sys

_root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "isaac_rover_2.0_amd")
_spec = importlib.util.spec_from_file_location(
    "isaac_rover_amd", os.path.join(_root, "__init__.py"), submodule_search_locations=[_root])
_pkg = importlib.util.module_from_spec(_spec)
sys.modules["isaac_rover_amd"] = _pkg
_spec.loader.exec_module(_pkg)

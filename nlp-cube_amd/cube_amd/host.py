"""Loader of `_cube_host` (csrc/host_collate.c): the bulk copies of the host-side featurisation.
The module is optional: without it the same arrays are built with NumPy alone (slower, identical)."""
import importlib.util
import os

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "_cube_host.so")
_mod = None


def load():
    """-> the extension module, or None when it has not been built.  A miss is not remembered: the
    file may be built later in the same process (`__graft_entry__.build()` on a clean tree)."""
    global _mod
    if _mod is None and os.path.exists(_PATH):
        spec = importlib.util.spec_from_file_location("_cube_host", _PATH)
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        _mod = m
    return _mod

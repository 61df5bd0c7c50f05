"""`import openEMS` shim: with <repo root> and <repo root>/fdtd-solver-antennas_amd/compat on sys.path the
reference's ``from openEMS import openEMS`` (antenna_sim/solver_fdtd_openems_fixed.py:132) resolves to
the MI355X backend.  See INTEGRATION.md §A."""
import importlib as _il
import os as _os
import sys as _sys

_root = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))))
if _root not in _sys.path:
    _sys.path.insert(0, _root)
_api = _il.import_module("fdtd-solver-antennas_amd.openems_api")
openEMS = _api.openEMS
from . import physical_constants  # noqa: E402,F401
import CSXCAD  # noqa: E402,F401  (the reference also does `from openEMS import CSXCAD`, fixed.py:100)

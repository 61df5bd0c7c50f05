"""`import openEMS` shim: put .../fdtd-solver-antennas_amd/compat on sys.path and the reference's
``from openEMS import openEMS`` (antenna_sim/solver_fdtd_openems_fixed.py:132) resolves to the
MI355X backend.  See INTEGRATION.md."""
import importlib as _il

_api = _il.import_module("fdtd-solver-antennas_amd.openems_api")
openEMS = _api.openEMS
from . import physical_constants  # noqa: E402,F401
from .. import CSXCAD  # noqa: E402,F401  (the reference also does `from openEMS import CSXCAD`, fixed.py:100)

"""`from openEMS.physical_constants import C0, EPS0` (antenna_sim/solver_fdtd_openems_fixed.py:133)."""
import importlib as _il

_c = _il.import_module("fdtd-solver-antennas_amd.constants")
C0 = _c.C0
MUE0 = _c.MU0
EPS0 = _c.EPS0
Z0 = _c.ETA0

"""Import alias: ``import fdtd_solver_antennas_amd`` -> the package in ./fdtd-solver-antennas_amd/."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("fdtd-solver-antennas_amd")
sys.modules[__name__] = _pkg

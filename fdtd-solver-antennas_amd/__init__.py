"""MI355X-native 3-D EC-FDTD backend behind the solver_fdtd_* plugin surface of
Veeryan/FDTD-solver-antennas.  The time-stepping hot path lives in csrc/ (hand-written HIP for
gfx950, C ABI in include/fdtd_hip.h); this package is the Python host side above it.

The directory name carries a hyphen (build contract), so import it with
``importlib.import_module("fdtd-solver-antennas_amd")`` or through the alias module
``fdtd_solver_antennas_amd`` at the repo root.
"""
__version__ = "0.1.0"

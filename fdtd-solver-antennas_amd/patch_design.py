"""Closed-form rectangular-patch design (Hammerstad) and 50-ohm microstrip width (Wheeler/Wadell).

Own implementation of the two helpers every ``prepare_*`` of the reference calls:
``design_patch_for_frequency`` (antenna_sim/physics.py:41-48, with effective_eps :19-28 and
delta_L :31-38) and ``calculate_microstrip_width`` (antenna_sim/solver_fdtd_openems_microstrip.py
:84-112).  Pinned against values obtained by importing the reference (tests/golden/design_values.json).
"""
from __future__ import annotations

import math

from .constants import C0


def effective_eps(eps_r: float, h_m: float, W_m: float) -> float:
    """Quasi-static effective permittivity of a microstrip of width W on height h."""
    if min(W_m, h_m) <= 0.0:
        return eps_r
    return 0.5 * (eps_r + 1.0) + 0.5 * (eps_r - 1.0) / math.sqrt(1.0 + 12.0 * h_m / W_m)


def delta_L(eps_eff: float, h_m: float, W_m: float) -> float:
    """Open-end length extension of the patch's radiating edge."""
    if min(W_m, h_m) <= 0.0:
        return 0.0
    r = W_m / h_m
    return 0.412 * h_m * ((eps_eff + 0.3) * (r + 0.264)) / ((eps_eff - 0.258) * (r + 0.8))


def design_patch_for_frequency(f_hz: float, eps_r: float, h_m: float):
    """(L, W, eps_eff) of a TM10 patch resonant at f_hz."""
    half_wave = C0 / (2.0 * f_hz)
    W = half_wave * math.sqrt(2.0 / (eps_r + 1.0))
    ee = effective_eps(eps_r, h_m, W)
    L = half_wave / math.sqrt(ee) - 2.0 * delta_L(ee, h_m, W)
    return L, W, ee


def calculate_microstrip_width(freq_hz: float, eps_r: float, h_m: float, z0: float = 50.0) -> float:
    """Strip width for characteristic impedance z0 (frequency is unused, as upstream)."""
    if z0 < 44.0:
        A = z0 / 60.0 * math.sqrt(0.5 * (eps_r + 1.0)) + (eps_r - 1.0) / (eps_r + 1.0) * (0.23 + 0.11 / eps_r)
        ratio = 8.0 * math.exp(A) / (math.exp(2.0 * A) - 2.0)
    else:
        B = 377.0 * math.pi / (2.0 * z0 * math.sqrt(eps_r))
        ratio = 2.0 / math.pi * (B - 1.0 - math.log(2.0 * B - 1.0)
                                 + (eps_r - 1.0) / (2.0 * eps_r) * (math.log(B - 1.0) + 0.39 - 0.61 / eps_r))
    return ratio * h_m

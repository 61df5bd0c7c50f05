"""Synthetic, deterministic benchmark/parity workloads.

All of them are the reference's primary scene — ``prepare_openems_patch_fixed``
(antenna_sim/solver_fdtd_openems_fixed.py:113-254): designed patch on a 60 x 60 mm FR-4 substrate,
ground plane, z-directed 50-ohm lumped port at x = -6 mm — put on a uniform-in-x/y grid with exactly
the node counts BASELINE.json names, 4 cells across the substrate and a graded z mesh.
"""
from __future__ import annotations

from dataclasses import dataclass
import numpy as np

from .grid import RectGrid
from .scene import Scene, voxelize
from .patch_design import design_patch_for_frequency
from .constants import EPS0


WORKLOADS = {
    # name: (nx, ny, nz, f0, steps quoted in BASELINE.md)
    "C2": (200, 200, 40, 2.45e9, 10000),
    "NS": (300, 300, 60, 2.45e9, 10000),
    "C3": (400, 400, 80, 2.45e9, 5000),
    "C4": (512, 512, 128, 5.8e9, 5000),
    "C5": (800, 800, 120, 2.45e9, 2000),
}


def graded_z_lines(nz: int, h: float, n_sub: int, dz_max: float, ratio: float = 1.3,
                   frac_below: float = 1.0 / 3.0) -> np.ndarray:
    """nz node lines: n_sub cells across the substrate [0, h], cells growing by `ratio` up to
    dz_max away from it, about frac_below of the remaining lines below the ground plane."""
    rest = nz - (n_sub + 1)
    if rest < 2:
        raise ValueError("nz too small")
    n_below = max(1, int(round(rest * frac_below)))
    n_above = rest - n_below
    dz0 = h / n_sub

    def steps(n):
        out, d = [], dz0
        for _ in range(n):
            d = min(d * ratio, dz_max)
            out.append(d)
        return np.array(out)

    below = -np.cumsum(steps(n_below))[::-1]
    above = h + np.cumsum(steps(n_above))
    return np.concatenate([below, np.linspace(0.0, h, n_sub + 1), above])


@dataclass
class PatchWorkload:
    name: str
    grid: RectGrid
    scene: Scene
    f0: float
    fc: float
    steps: int


def patch_workload(name: str = "NS", *, nx=None, ny=None, nz=None, f0=None, eps_r=4.3, h=1.6e-3,
                   loss_tangent=0.02, box_xy=0.2) -> PatchWorkload:
    """The `fixed` patch scene on an nx x ny x nz node grid (uniform in x/y over `box_xy` metres)."""
    if name in WORKLOADS:
        wnx, wny, wnz, wf0, steps = WORKLOADS[name]
    else:
        wnx, wny, wnz, wf0, steps = 80, 80, 30, 2.45e9, 1000
    nx, ny, nz, f0 = nx or wnx, ny or wny, nz or wnz, f0 or wf0
    x = np.linspace(-box_xy / 2, box_xy / 2, nx)
    y = np.linspace(-box_xy / 2, box_xy / 2, ny)
    dxy = float(x[1] - x[0])
    z = graded_z_lines(nz, h, 4, dz_max=max(dxy, h / 4))
    grid = RectGrid(x, y, z)
    L, W, _ = design_patch_for_frequency(f0, eps_r, h)
    u = 1e-3
    sc = Scene(unit=u)
    pw, pl, hh = W / u, L / u, h / u          # fixed.py:143-149 puts W on x, L on y
    sc.add_metal("patch").add_box([-pw / 2, -pl / 2, hh], [pw / 2, pl / 2, hh], priority=10)
    kappa = 2 * np.pi * f0 * EPS0 * eps_r * loss_tangent
    sc.add_material("substrate", eps_r, kappa).add_box([-30, -30, 0], [30, 30, hh], priority=0)
    sc.add_metal("gnd").add_box([-30, -30, 0], [30, 30, 0], priority=10)
    sc.add_lumped_port(1, 50.0, [-6, 0, 0], [-6, 0, hh], "z", 1.0, priority=5)
    return PatchWorkload(name, grid, sc, f0, f0 / 2.0, steps)

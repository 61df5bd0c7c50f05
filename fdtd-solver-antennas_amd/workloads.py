"""Synthetic, deterministic benchmark/parity workloads: the reference's
scenes put on a uniform-in-x/y grid with EXACTLY the node counts BASELINE.json names ("uniform-grid
override"), 4 cells across the substrate and a graded z mesh.

  C2 / NS / C3 : ``prepare_openems_patch_fixed`` scene (antenna_sim/solver_fdtd_openems_fixed.py:113-254):
                 designed 2.45 GHz patch on a 60 x 60 mm FR-4 substrate, ground, lumped port at x = -6 mm
  C4           : 5.8 GHz ``prepare_openems_microstrip_patch_3d`` geometry (solver_fdtd_openems_microstrip_3d.py:19-196)
  C5           : 2 x 2 array at 61.2 mm pitch through ``prepare_openems_microstrip_multi_3d``
                 (solver_fdtd_openems_microstrip_multi_3d.py:98-593), four lumped ports excited in phase
C4/C5 reuse this package's own scene builders (solver_fdtd_hip.prepare_hip_*) and only replace the mesh.
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
    # (BASELINE config 2 has 40 planes for two 10-cell CPML layers: with a third of the spare planes below the ground plane, as on the other
    # grids, it would sit on plane 12 — the very plane of the NF2FF box's lower face (layer + 2) — and the far field would be that of a
    # surface cut along a PEC sheet (D = 14.6 dBi); 14 planes below put the face two cells under the ground plane)
    z = graded_z_lines(nz, h, 4, dz_max=max(dxy, h / 4), frac_below=0.41 if name == "C2" else 1.0 / 3.0)
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


def _uniform_lines(lo: float, hi: float, n: int) -> np.ndarray:
    return np.linspace(lo, hi, n)


def workload_from_prepared(name: str, prepared, nx: int, ny: int, nz: int, steps: int) -> PatchWorkload:
    """Take the scene a prepare_hip_* call drew (boxes, materials, ports, simulation box) and put it on an
    nx x ny x nz grid: uniform x/y over the simulation box, z = 4 cells across the (first) substrate and
    graded away from it.  Metal sheets / thin copper snap to the grid through the voxeliser."""
    from .scene import Scene, Box as SceneBox
    fdtd = prepared.FDTD
    csx = fdtd.GetCSX()
    unit = csx.GetGrid().GetDeltaUnit()
    ext = [np.asarray(csx.GetGrid().GetLines(a), float) * unit for a in range(3)]
    x = _uniform_lines(ext[0].min(), ext[0].max(), nx)
    y = _uniform_lines(ext[1].min(), ext[1].max(), ny)
    subs = [b for p in csx.properties if p.kind == "Material" for b in p.boxes]
    zs = sorted({round(float(v), 9) for b in subs for v in ((b.matrix @ np.append(b.start, 1.0))[2] * unit,
                                                            (b.matrix @ np.append(b.stop, 1.0))[2] * unit)})
    z_lo, z_hi = zs[0], zs[-1]
    h = z_hi - z_lo
    dxy = float(x[1] - x[0])
    frac_below = (z_lo - ext[2].min()) / max((z_lo - ext[2].min()) + (ext[2].max() - z_hi), 1e-30)
    z = graded_z_lines(nz, h, 4, dz_max=max(dxy, h / 4), frac_below=float(np.clip(frac_below, 0.2, 0.5))) + z_lo
    grid = RectGrid(x, y, z)
    sc = Scene(unit=unit)
    for p in csx.properties:
        tgt = sc.add_material(p.name, p.params.get("epsilon", 1.0), p.params.get("kappa", 0.0)) if p.kind == "Material" \
            else sc.add_metal(p.name)
        for b in p.boxes:
            tgt.boxes.append(SceneBox(tuple(b.start), tuple(b.stop), b.priority, b.matrix.copy()))
    for port in fdtd._ports:
        sc.add_lumped_port(port.number, port.R, port.start, port.stop, port.exc_ny, port.excite, port.priority)
    return PatchWorkload(name, grid, sc, fdtd._f0, fdtd._fc, steps)


def baseline_workload(name: str) -> PatchWorkload:
    """The workload BASELINE.md names for `name` (C2, NS, C3: fixed scene; C4: 5.8 GHz microstrip-3D; C5: 2x2 array)."""
    from . import solver_fdtd_hip as s
    from .params import PatchAntennaParams
    nx, ny, nz, f0, steps = WORKLOADS[name]
    if name == "C4":
        p = PatchAntennaParams.from_user_units(frequency_ghz=5.8, er=4.3, h_mm=1.6, loss_tangent=0.02)
        prep = s.prepare_hip_microstrip_patch_3d(p, feed_direction=s.FeedDirection.NEG_X, boundary="PML_8")
        if not prep.ok:
            raise RuntimeError(prep.message)
        return workload_from_prepared(name, prep, nx, ny, nz, steps)
    if name == "C5":
        p = PatchAntennaParams.from_user_units(frequency_ghz=2.45, er=4.3, h_mm=1.6, loss_tangent=0.02)
        pitch = 0.0612
        arr = [s.PatchInstance(f"P{n}", p, (ix - 0.5) * pitch, (iy - 0.5) * pitch, 0.0, s.FeedDirection.NEG_X)
               for n, (ix, iy) in enumerate([(0, 0), (1, 0), (0, 1), (1, 1)])]
        prep = s.prepare_hip_microstrip_multi_3d(arr, boundary="PML_8")
        if not prep.ok:
            raise RuntimeError(prep.message)
        return workload_from_prepared(name, prep, nx, ny, nz, steps)
    return patch_workload(name)

"""Convolutional PML (Roden & Gedney 2000) profiles for the EC update.

The reference asks its external engine for 'PML_8' / [3]*6 boundaries
(antenna_sim/solver_fdtd_openems_microstrip_3d.py:84, solver_fdtd_openems.py:188, gui default
gui_app.py:190); that engine implements a split-field UPML.  The north-star mandates CPML
psi-field layers instead, so this is a NEW formulation, not a restatement: in EC form every
difference d taken along a PML axis becomes

    psi <- b*psi + c*d ,   d <- d/kappa + psi          (include/fdtd_hip.h, fdtd_set_cpml)

with b = exp(-(sigma/kappa + alpha) dt/eps0), c = sigma (b-1) / (kappa (sigma + kappa alpha)).
Stretched-coordinate PML is material independent, so eps0 is used throughout.
"""
from __future__ import annotations

from dataclasses import dataclass
import numpy as np

from .constants import EPS0, ETA0
from .grid import RectGrid


@dataclass
class CPMLSpec:
    cells: tuple = (10, 10, 10, 10, 10, 10)   # x-,x+,y-,y+,z-,z+ ; 0 disables a face
    order: float = 3.0
    sigma_factor: float = 1.0                 # x sigma_opt = 0.8 (m+1) / (eta0 * delta)
    kappa_max: float = 1.0
    alpha_max: float = 0.05                   # S/m, linearly decreasing into the layer


@dataclass
class CPMLTables:
    slot: list            # per axis int32 [n_a], -1 outside the layers
    nslot: list
    coef: list            # per axis float32 [2 (E-loc, H-loc)][3 (b, c, 1/kappa)][n_a]

    def for_slab(self, k0: int, nk: int, dtype=np.float32):
        """Slot maps + packed coefficient block for a z-slab [k0, k0+nk) (C ABI layout)."""
        sz = self.slot[2][k0:k0 + nk].copy()
        own = sz >= 0
        sz[own] = np.arange(int(own.sum()), dtype=np.int32)
        coef = np.concatenate([self.coef[0].ravel(), self.coef[1].ravel(),
                               np.ascontiguousarray(self.coef[2][:, :, k0:k0 + nk]).ravel()]).astype(dtype)
        return (self.slot[0], self.slot[1], sz, self.nslot[0], self.nslot[1], int(own.sum()), coef)


def build_cpml(grid: RectGrid, dt: float, spec: CPMLSpec, dtype=np.float32) -> CPMLTables:
    slots, nslots, coefs = [], [], []
    for a in range(3):
        l = grid.lines[a]
        n = l.size
        lo, hi = int(spec.cells[2 * a]), int(spec.cells[2 * a + 1])
        if lo + hi > n - 2:
            raise ValueError(f"CPML thicker than the grid on axis {a}")
        slot = -np.ones(n, np.int32)
        slot[:lo] = np.arange(lo)
        if hi:
            # through the last line (inert there: b = c = 0) so the range is anchored at the end
            slot[n - 1 - hi:] = lo + np.arange(hi + 1)
        coef = np.zeros((2, 3, n), np.float64)
        coef[:, 2, :] = 1.0
        pos = [l, np.append(0.5 * (l[:-1] + l[1:]), l[-1])]   # E-located nodes, H-located half nodes
        for eh in range(2):
            p = pos[eh]
            rho = np.zeros(n)
            delta = np.ones(n)
            if lo:
                thick = l[lo] - l[0]
                r = (l[lo] - p) / thick
                msk = r > 0
                rho[msk] = r[msk]
                delta[msk] = thick / lo
            if hi:
                thick = l[-1] - l[n - 1 - hi]
                r = (p - l[n - 1 - hi]) / thick
                msk = r > 0
                rho[msk] = r[msk]
                delta[msk] = thick / hi
            rho = np.clip(rho, 0.0, 1.0)
            act = rho > 0
            if eh == 0:
                act[0] = act[-1] = False     # boundary nodes: PEC-backed, no update there
            else:
                act[-1] = False              # there is no half node beyond the last line
            sig = spec.sigma_factor * 0.8 * (spec.order + 1.0) / (ETA0 * delta) * rho ** spec.order
            kap = 1.0 + (spec.kappa_max - 1.0) * rho ** spec.order
            alp = spec.alpha_max * (1.0 - rho)
            b = np.exp(-(sig / kap + alp) * dt / EPS0)
            with np.errstate(divide="ignore", invalid="ignore"):
                c = np.where(sig > 0, sig * (b - 1.0) / (kap * (sig + kap * alp)), 0.0)
            coef[eh, 0, act] = b[act]
            coef[eh, 1, act] = c[act]
            coef[eh, 2, act] = 1.0 / kap[act]
        slots.append(slot)
        nslots.append(lo + hi + (1 if hi else 0))
        coefs.append(coef.astype(dtype))
    return CPMLTables(slot=slots, nslot=nslots, coef=coefs)

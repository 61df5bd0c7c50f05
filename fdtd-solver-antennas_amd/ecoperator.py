"""Equivalent-circuit (EC) FDTD operator: per-edge update coefficients from materials + mesh.

Replaces the operator set-up phase that the reference triggers inside ``FDTD.Run(...)``
(antenna_sim/solver_fdtd_openems_fixed.py:280).  [EXT] follows the published
EC-FDTD formulation openEMS implements:

    C = eps_eff * A~/l      G = kappa_eff * A~/l (+ lumped G)      L = mu0 * A/l~
    vv = (1 - dt G/2C)/(1 + dt G/2C)     vi = (dt/C)/(1 + dt G/2C)      ii = 1     iv = dt/L

with eps_eff/kappa_eff the area-weighted mean over the (up to) four cells around an edge.
All arrays are [3][nz][ny][nx] (component, then z, y, x; x fastest) — the host layout of
include/fdtd_hip.h.

Two output forms:
  * raw     — the four coefficient arrays as float32 (12 B/cell-step of extra HBM traffic ×4);
  * classes — one uint8 class per edge + separable 1-D metric tables (3 B/cell-step), exactly
              the factorisation documented at fdtd_set_operator_classes().

This numpy formulation is the SPEC of the operator.  The product path does not run it: Simulation.build hands
materials + mesh to fdtd_build_operator (HIP kernels in csrc/opbuild.hip; plain C in the oracle), which must
reproduce these arrays bit for bit (tests/test_operator_build_*.py); only metric_lists / lumped_overrides (1-D
tables, a handful of edges) are evaluated on the host.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Sequence
import numpy as np

from .constants import EPS0, MU0
from .grid import RectGrid


@dataclass
class LumpedEdge:
    comp: int
    i: int
    j: int
    k: int
    G: float  # conductance added to this edge [S]


@dataclass
class ECOperator:
    grid: RectGrid
    dt: float
    # per-edge class factors (float64, before compression): vv and m with vi = m * l / A~
    vv: np.ndarray          # [3][nz][ny][nx] float32
    m: np.ndarray           # [3][nz][ny][nx] float32
    emet: list              # emet[c][axis] -> 1-D float32 table
    hmet: list              # hmet[c][axis] -> 1-D float32 table

    # ---- raw form ------------------------------------------------------------------------
    def raw(self, k0: int = 0, nk: Optional[int] = None):
        """(vv, vi, ii, iv) float32 [3][nk][ny][nx], expanded with the float32 association the
        C ABI fixes, so raw and class forms agree bit for bit."""
        nx, ny, nz = self.grid.shape
        nk = nz - k0 if nk is None else nk
        sl = slice(k0, k0 + nk)
        vi = np.empty((3, nk, ny, nx), np.float32)
        iv = np.empty((3, nk, ny, nx), np.float32)
        for c in range(3):
            ex, ey, ez = self.emet[c]
            hx, hy, hz = self.hmet[c]
            eyz = (ey[None, :, None] * ez[sl, None, None]).astype(np.float32)
            vi[c] = self.m[c, sl] * (ex[None, None, :] * eyz)
            hyz = (hy[None, :, None] * hz[sl, None, None]).astype(np.float32)
            iv[c] = hx[None, None, :] * hyz
        vv = np.ascontiguousarray(self.vv[:, sl])
        ii = np.ones_like(vv)
        return vv, vi, ii, iv

    # ---- class form ----------------------------------------------------------------------
    def classes(self, k0: int = 0, nk: Optional[int] = None):
        """(ecls uint8 [3][nk][ny][nx], cls_vv, cls_m) or None if more than 256 classes."""
        nx, ny, nz = self.grid.shape
        nk = nz - k0 if nk is None else nk
        sl = slice(k0, k0 + nk)
        key = np.empty((3, nk, ny, nx), np.uint64)
        key[...] = np.ascontiguousarray(self.vv[:, sl]).view(np.uint32).astype(np.uint64) << np.uint64(32)
        key |= np.ascontiguousarray(self.m[:, sl]).view(np.uint32).astype(np.uint64)
        flat = key.ravel()
        try:                               # hash-based, O(N): ~10x faster than the sort in np.unique at 1e8 edges
            import pandas as pd
            inv, uniq = pd.factorize(flat)
            uniq = np.asarray(uniq, dtype=np.uint64)
        except ImportError:
            uniq, inv = np.unique(flat, return_inverse=True)
        if uniq.size > 256:
            return None
        cls_vv = (uniq >> np.uint64(32)).astype(np.uint32).view(np.float32)
        cls_m = (uniq & np.uint64(0xFFFFFFFF)).astype(np.uint32).view(np.float32)
        return inv.astype(np.uint8).reshape(3, nk, ny, nx), cls_vv.copy(), cls_m.copy()

    def metric_tables(self, k0: int = 0, nk: Optional[int] = None):
        """emet, hmet packed as [3][nx + ny + nk] float32 (z part local to the slab)."""
        return pack_metric_tables(self.emet, self.hmet, self.grid, k0, nk)


def _edge_average(cellval: np.ndarray, grid: RectGrid, comp: int) -> np.ndarray:
    """Area-weighted mean of a cell quantity [nz-1][ny-1][nx-1] over the cells around every
    comp-directed edge -> [nz][ny][nx] (non-existent edges get the value of a neighbour)."""
    nx, ny, nz = grid.shape
    d = [grid.d[a][:-1] for a in range(3)]  # true cell sizes
    axes_np = {0: 2, 1: 1, 2: 0}            # physical axis -> numpy axis
    a1, a2 = (comp + 1) % 3, (comp + 2) % 3
    w_shape = [1, 1, 1]
    w_shape[axes_np[a1]] = d[a1].size
    w1 = d[a1].reshape(w_shape)
    w_shape = [1, 1, 1]
    w_shape[axes_np[a2]] = d[a2].size
    w2 = d[a2].reshape(w_shape)
    w = w1 * w2                                   # broadcastable quarter-area weights (x4)
    num = cellval * w
    den = np.broadcast_to(w, cellval.shape)
    pad = [(0, 0)] * 3
    pad[axes_np[a1]] = (1, 1)
    pad[axes_np[a2]] = (1, 1)
    num = np.pad(num, pad)
    den = np.pad(den, pad)

    def sum4(p):
        s1 = [slice(None)] * 3
        out = 0.0
        for o1 in (0, 1):
            for o2 in (0, 1):
                s = list(s1)
                n1 = p.shape[axes_np[a1]] - 1
                n2 = p.shape[axes_np[a2]] - 1
                s[axes_np[a1]] = slice(o1, o1 + n1)
                s[axes_np[a2]] = slice(o2, o2 + n2)
                out = out + p[tuple(s)]
        return out

    avg = sum4(num) / sum4(den)                    # shape: n along a1,a2 ; n-1 along comp
    pad = [(0, 0)] * 3
    pad[axes_np[comp]] = (0, 1)
    return np.pad(avg, pad, mode="edge")


def build_operator(grid: RectGrid, eps_r: np.ndarray, kappa: np.ndarray, pec: np.ndarray,
                   dt: float, lumped: Sequence[LumpedEdge] = ()) -> ECOperator:
    """eps_r, kappa: per cell [nz-1][ny-1][nx-1]; pec: bool [3][nz][ny][nx] (True = PEC edge)."""
    nx, ny, nz = grid.shape
    if eps_r.shape != (nz - 1, ny - 1, nx - 1) or kappa.shape != eps_r.shape:
        raise ValueError("cell arrays must be [nz-1][ny-1][nx-1]")
    if pec.shape != (3, nz, ny, nx):
        raise ValueError("pec must be [3][nz][ny][nx]")
    n_axis = (nx, ny, nz)
    vv = np.empty((3, nz, ny, nx), np.float32)
    m = np.empty((3, nz, ny, nx), np.float32)
    for c in range(3):
        eps_e = _edge_average(eps_r, grid, c) * EPS0
        kap_e = _edge_average(kappa, grid, c)
        x = 0.5 * dt * kap_e / eps_e
        vv_c = (1.0 - x) / (1.0 + x)
        m_c = dt / (eps_e * (1.0 + x))
        dead = pec[c].copy()
        # non-existent last edges + tangential edges on the six outer planes (PEC backing; Mur
        # overwrites them after the update anyway)
        idx = [slice(None)] * 3
        np_axis = {0: 2, 1: 1, 2: 0}
        idx[np_axis[c]] = -1
        dead[tuple(idx)] = True
        for a in range(3):
            if a == c:
                continue
            for side in (0, -1):
                idx = [slice(None)] * 3
                idx[np_axis[a]] = side
                dead[tuple(idx)] = True
        vv_c[dead] = 0.0
        m_c[dead] = 0.0
        vv[c] = vv_c
        m[c] = m_c
    emet, hmet = metric_lists(grid, dt)
    op = ECOperator(grid=grid, dt=dt, vv=vv, m=m, emet=emet, hmet=hmet)
    # lumped conductances: (vv, m) of those edges with G_total = kappa*A~/l + G
    edge, comp, o_vv, o_m = lumped_overrides(grid, eps_r, kappa, pec, dt, lumped)
    op.vv.reshape(3, -1)[comp, edge] = o_vv
    op.m.reshape(3, -1)[comp, edge] = o_m
    return op


def metric_lists(grid: RectGrid, dt: float, dtype=np.float32):
    """Separable metric of the EC operator: vi = m * l[c] / (dd[a1] * dd[a2]);  iv = (dt/mu0) * dd[c] / (d[a1] * d[a2]).
    emet[c][axis], hmet[c][axis] -> 1-D float32 tables over the whole grid (`dtype`: the C ABI takes float32; float64 is what
    the double-precision checker of the fp32 error budget is handed, tests/helpers.py)."""
    emet, hmet = [], []
    for c in range(3):
        et, ht = [None] * 3, [None] * 3
        for a in range(3):
            if a == c:
                l = grid.d[a].copy()
                l[-1] = 0.0
                et[a] = l.astype(dtype)
                ht[a] = (dt / MU0 * grid.dd[a]).astype(dtype)
            else:
                et[a] = (1.0 / grid.dd[a]).astype(dtype)
                inv = 1.0 / grid.d[a]
                inv[-1] = 0.0
                ht[a] = inv.astype(dtype)
        emet.append(et)
        hmet.append(ht)
    return emet, hmet


def pack_metric_tables(emet, hmet, grid: RectGrid, k0: int = 0, nk: Optional[int] = None, dtype=np.float32):
    """emet, hmet packed as [3][nx + ny + nk] float32 (z part local to the slab) — the C ABI's table argument."""
    nx, ny, nz = grid.shape
    nk = nz - k0 if nk is None else nk
    sl = slice(k0, k0 + nk)
    e = np.concatenate([np.concatenate([t[0], t[1], t[2][sl]]) for t in emet]).astype(dtype)
    h = np.concatenate([np.concatenate([t[0], t[1], t[2][sl]]) for t in hmet]).astype(dtype)
    return e.reshape(3, nx + ny + nk), h.reshape(3, nx + ny + nk)


def lumped_overrides(grid: RectGrid, eps_r: np.ndarray, kappa: np.ndarray, pec: np.ndarray, dt: float,
                     lumped: Sequence[LumpedEdge], dtype=np.float32):
    """(global edge index int64, comp int8, vv float32, m float32) of the edges that carry a lumped conductance —
    the few coefficients the host fixes itself when the operator is built on the device (fdtd_build_operator)."""
    nx, ny, nz = grid.shape
    edge, comp, o_vv, o_m = [], [], [], []
    for le in lumped:
        c, i, j, k = le.comp, le.i, le.j, le.k
        if pec[c, k, j, i]:
            continue
        pos = (i, j, k)
        a1, a2 = (c + 1) % 3, (c + 2) % 3
        l = grid.d[c][pos[c]]
        A = grid.dd[a1][pos[a1]] * grid.dd[a2][pos[a2]]
        eps_e = _edge_scalar(eps_r, grid, c, pos) * EPS0
        kap_e = _edge_scalar(kappa, grid, c, pos)
        C = eps_e * A / l
        G = kap_e * A / l + le.G
        x = 0.5 * dt * G / C
        edge.append((k * ny + j) * nx + i)
        comp.append(c)
        o_vv.append((1.0 - x) / (1.0 + x))
        o_m.append(dt / (eps_e * (1.0 + x)))
    return (np.asarray(edge, np.int64), np.asarray(comp, np.int8), np.asarray(o_vv, dtype), np.asarray(o_m, dtype))


def _edge_scalar(cellval, grid, comp, pos):
    """Same average as _edge_average for a single edge (used for the few lumped edges)."""
    nx, ny, nz = grid.shape
    ncell = (nx - 1, ny - 1, nz - 1)
    a1, a2 = (comp + 1) % 3, (comp + 2) % 3
    num = den = 0.0
    for o1 in (-1, 0):
        for o2 in (-1, 0):
            cidx = list(pos)
            cidx[a1] += o1
            cidx[a2] += o2
            if not (0 <= cidx[a1] < ncell[a1] and 0 <= cidx[a2] < ncell[a2] and 0 <= cidx[comp] < ncell[comp]):
                continue
            w = grid.d[a1][cidx[a1]] * grid.d[a2][cidx[a2]]
            num += w * float(cellval[cidx[2], cidx[1], cidx[0]])
            den += w
    return num / den

"""Near-field recording surfaces + near-to-far-field transform.

Replaces ``FDTD.CreateNF2FFBox()`` / ``nf2ff.CalcNF2FF(sim_path, f, theta, phi, center=...)``
(antenna_sim/solver_fdtd_openems_fixed.py:220,296; per-phi loops solver_fdtd_openems_microstrip_3d.py
:224-225 and _multi_3d.py:620-621).  The external engine dumps time-domain E/H on six faces to HDF5
and a separate tool DFTs and integrates them; here the RAW edge voltages / face currents of the faces
are either recorded in the time domain in HBM and transformed on the device for any frequency after the
run (fdtd_set_recorder / fdtd_rec_transform: the reference's semantics, default) or accumulated as a
running DFT at frequencies fixed beforehand (fdtd_set_dft: constant memory, for very long runs);
interpolation to the face nodes, the equivalent currents and the radiation integral (fdtd_farfield, on
the GPU) happen once at the end.

Result attributes mirror what the reference reads from the openEMS result object:
``E_norm[f]`` -> (ntheta, nphi), ``Dmax[f]`` (fixed.py:304-305) and ``E_theta, E_phi, P_rad, Prad``
(solver_fdtd_openems.py:318-321).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple
import numpy as np

from .constants import C0, ETA0
from .grid import RectGrid
from ._capi import KIND_V, KIND_I


@dataclass
class _BoxReq:
    kind: int
    comp: int
    lo: Tuple[int, int, int]
    hi: Tuple[int, int, int]


class NF2FFBox:
    """Six recording faces on node planes lo[a] / hi[a] (inclusive node indices)."""

    def __init__(self, grid: RectGrid, lo: Sequence[int], hi: Sequence[int]):
        self.grid = grid
        self.lo = tuple(int(v) for v in lo)
        self.hi = tuple(int(v) for v in hi)
        n = grid.shape
        for a in range(3):
            if not (1 <= self.lo[a] < self.hi[a] <= n[a] - 2):
                raise ValueError("NF2FF box must lie strictly inside the grid")
        self.requests: List[_BoxReq] = []
        self._req_of: Dict[Tuple[int, int, int], int] = {}   # (face, kind, comp) -> request index
        for f in range(6):
            a, side = f // 2, f % 2
            a0 = self.hi[a] if side else self.lo[a]
            t1, t2 = (a + 1) % 3, (a + 2) % 3
            for t, u in ((t1, t2), (t2, t1)):
                lo_v, hi_v = list(self.lo), list(self.hi)
                lo_v[a] = hi_v[a] = a0
                lo_v[t] -= 1                     # E_t at a node averages edges t-1 and t
                self._add(f, KIND_V, t, lo_v, hi_v)
                lo_i, hi_i = list(self.lo), list(self.hi)
                lo_i[a], hi_i[a] = a0 - 1, a0    # H_t at a node averages the 4 faces around it
                lo_i[u] -= 1
                self._add(f, KIND_I, t, lo_i, hi_i)

    def _add(self, face, kind, comp, lo, hi):
        self._req_of[(face, kind, comp)] = len(self.requests)
        self.requests.append(_BoxReq(kind, comp, tuple(lo), tuple(hi)))

    # -- recording ------------------------------------------------------------------------------
    def register(self, engine) -> List[int]:
        return [engine.add_dft_box(r.kind, r.comp, r.lo, r.hi) for r in self.requests]

    def collect(self, engine, ids: List[int], tw_v=None, tw_i=None) -> List[np.ndarray]:
        """Per request a complex array [nfreq][nk][nj][ni] covering the whole request box, zero where
        this slab owns nothing (sum over ranks = complete box).  With twiddle tables (recorder mode) the
        recorded time-domain samples are transformed on the device for those frequencies."""
        out = []
        for r, bid in zip(self.requests, ids):
            ext = [r.hi[a] - r.lo[a] + 1 for a in range(3)]
            if tw_v is not None:
                data, lo, hi = engine.rec_transform(bid, tw_v if r.kind == KIND_V else tw_i)
            else:
                data, lo, hi = engine.get_dft_box(bid)
            full = np.zeros((data.shape[0], ext[2], ext[1], ext[0]), np.complex128)
            if data.size:
                full[:, lo[2] - r.lo[2]:hi[2] - r.lo[2] + 1, lo[1] - r.lo[1]:hi[1] - r.lo[1] + 1,
                     lo[0] - r.lo[0]:hi[0] - r.lo[0] + 1] = data
            out.append(full)
        return out

    # -- post-processing ------------------------------------------------------------------------
    def surface_currents(self, boxes: List[np.ndarray], fidx: int, center_m: Sequence[float]):
        """Node-interpolated E,H on the six faces -> quadrature points.
        Returns pos [N][3] (relative to center), Js, Ms [N][3] complex (area weighted) and the
        outward Poynting flux 0.5*Re sum (E x H*).n dA."""
        g = self.grid
        pos_l, js_l, ms_l = [], [], []
        flux = 0.0
        np_axis = {0: 2, 1: 1, 2: 0}
        for f in range(6):
            a, side = f // 2, f % 2
            t1, t2 = (a + 1) % 3, (a + 2) % 3
            nsign = 1.0 if side else -1.0
            E = {}
            H = {}
            for t, u in ((t1, t2), (t2, t1)):
                v = boxes[self._req_of[(f, KIND_V, t)]][fidx]        # [k][j][i] over the request box
                # divide by edge length along t, then average edges (t-1, t)
                lt = g.d[t][self.lo[t] - 1:self.hi[t] + 1]
                shp = [1, 1, 1]; shp[np_axis[t]] = lt.size
                e = v / lt.reshape(shp)
                sl_a = [slice(None)] * 3; sl_b = [slice(None)] * 3
                sl_a[np_axis[t]] = slice(0, -1); sl_b[np_axis[t]] = slice(1, None)
                E[t] = np.squeeze(0.5 * (e[tuple(sl_a)] + e[tuple(sl_b)]), axis=np_axis[a])
                i_ = boxes[self._req_of[(f, KIND_I, t)]][fidx]
                ld = g.dd[t][self.lo[t]:self.hi[t] + 1]
                shp = [1, 1, 1]; shp[np_axis[t]] = ld.size
                h = i_ / ld.reshape(shp)
                # average over the two planes along a and the pair (u-1, u)
                acc = 0.0
                for oa in (0, 1):
                    for ou in (0, 1):
                        s = [slice(None)] * 3
                        s[np_axis[a]] = slice(oa, oa + 1)
                        s[np_axis[u]] = slice(ou, ou + (self.hi[u] - self.lo[u] + 1))
                        acc = acc + h[tuple(s)]
                H[t] = np.squeeze(0.25 * acc, axis=np_axis[a])
            # trapezoid weights along the two tangential axes
            w = {}
            for t in (t1, t2):
                x = g.lines[t][self.lo[t]:self.hi[t] + 1]
                wt = np.empty_like(x)
                wt[1:-1] = 0.5 * (x[2:] - x[:-2]); wt[0] = 0.5 * (x[1] - x[0]); wt[-1] = 0.5 * (x[-1] - x[-2])
                w[t] = wt
            # 2-D arrays are ordered (slow, fast) = (larger axis id, smaller axis id) after the squeeze
            slow, fast = (t1, t2) if t1 > t2 else (t2, t1)
            dA = w[slow][:, None] * w[fast][None, :]
            coords = [None] * 3
            coords[a] = np.full(dA.shape, g.lines[a][self.hi[a] if side else self.lo[a]])
            cs, cf = np.meshgrid(g.lines[slow][self.lo[slow]:self.hi[slow] + 1],
                                 g.lines[fast][self.lo[fast]:self.hi[fast] + 1], indexing="ij")
            coords[slow], coords[fast] = cs, cf
            Ev = [np.zeros(dA.shape, np.complex128) for _ in range(3)]
            Hv = [np.zeros(dA.shape, np.complex128) for _ in range(3)]
            Ev[t1], Ev[t2] = E[t1], E[t2]
            Hv[t1], Hv[t2] = H[t1], H[t2]
            n = [0.0, 0.0, 0.0]; n[a] = nsign
            # Js = n x H ; Ms = -n x E  (only the normal component of n is non-zero)
            J = [np.zeros(dA.shape, np.complex128) for _ in range(3)]
            M = [np.zeros(dA.shape, np.complex128) for _ in range(3)]
            J[t1] = -nsign * Hv[t2]; J[t2] = nsign * Hv[t1]       # (n x H)_{t1} = -n_a H_{t2}; (n x H)_{t2} = n_a H_{t1}
            M[t1] = nsign * Ev[t2]; M[t2] = -nsign * Ev[t1]
            # Poynting: (E x H*)_a = E_t1 H_t2* - E_t2 H_t1*
            S = Ev[t1] * np.conj(Hv[t2]) - Ev[t2] * np.conj(Hv[t1])
            flux += 0.5 * nsign * float(np.sum(np.real(S) * dA))
            pos_l.append(np.stack([coords[0].ravel() - center_m[0], coords[1].ravel() - center_m[1],
                                   coords[2].ravel() - center_m[2]], axis=1))
            js_l.append(np.stack([(J[c] * dA).ravel() for c in range(3)], axis=1))
            ms_l.append(np.stack([(M[c] * dA).ravel() for c in range(3)], axis=1))
        return np.concatenate(pos_l), np.concatenate(js_l), np.concatenate(ms_l), flux


@dataclass
class NF2FFResult:
    freq: np.ndarray
    theta: np.ndarray          # rad
    phi: np.ndarray            # rad
    E_theta: List[np.ndarray] = field(default_factory=list)   # per frequency (ntheta, nphi) complex, r*E
    E_phi: List[np.ndarray] = field(default_factory=list)
    E_norm: List[np.ndarray] = field(default_factory=list)
    P_rad: List[np.ndarray] = field(default_factory=list)     # radiation intensity [W/sr]
    Prad: List[float] = field(default_factory=list)           # total radiated power [W]
    Dmax: List[float] = field(default_factory=list)


def calc_nf2ff(lib, box: NF2FFBox, boxes: List[np.ndarray], freqs, theta_rad, phi_rad, center_m,
               device: int = 0) -> NF2FFResult:
    """The whole theta x phi grid in one GPU launch per frequency."""
    from ._capi import farfield
    freqs = np.atleast_1d(np.asarray(freqs, float))
    th = np.asarray(theta_rad, float); ph = np.asarray(phi_rad, float)
    TH, PH = np.meshgrid(th, ph, indexing="ij")
    res = NF2FFResult(freq=freqs, theta=th, phi=ph)
    for fi, f in enumerate(freqs):
        pos, Js, Ms, flux = box.surface_currents(boxes, fi, center_m)
        k = 2.0 * np.pi * f / C0
        eth, eph = farfield(lib, pos, Js, Ms, k, TH.ravel(), PH.ravel(), device=device)
        eth = eth.reshape(TH.shape); eph = eph.reshape(TH.shape)
        U = (np.abs(eth) ** 2 + np.abs(eph) ** 2) / (2.0 * ETA0)
        res.E_theta.append(eth); res.E_phi.append(eph)
        res.E_norm.append(np.sqrt(np.abs(eth) ** 2 + np.abs(eph) ** 2))
        res.P_rad.append(U)
        res.Prad.append(flux)
        res.Dmax.append(4.0 * np.pi * float(np.max(U)) / flux if flux > 0 else float("nan"))
    return res

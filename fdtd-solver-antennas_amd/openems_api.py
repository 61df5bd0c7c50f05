"""Host-side mirror of the openEMS / CSXCAD Python interface subset the reference drives.

The reference reaches its FDTD engine exclusively through these calls
(antenna_sim/solver_fdtd_openems_fixed.py:171-220,280,296 and the sibling solver files):

    FDTD = openEMS(NrTS=, EndCriteria=) ; FDTD.SetGaussExcite ; FDTD.SetBoundaryCond ; FDTD.SetCSX
    CSX = ContinuousStructure() ; mesh = CSX.GetGrid() ; mesh.SetDeltaUnit/AddLine/SmoothMeshLines
    CSX.AddMetal/AddMaterial(...).AddBox(...)[.AddTransform(...)]
    FDTD.AddEdges2Grid ; FDTD.AddLumpedPort ; FDTD.CreateNF2FFBox ; FDTD.Run
    nf2ff.CalcNF2FF(sim_path, f, theta_deg, phi_deg, center=) -> .E_norm[0], .Dmax[0], ...
    port.CalcPort(sim_path, f) -> .uf_inc, .uf_ref, ...        (microstrip.py:409-412)

Same names, argument meaning and result attributes — but ``Run`` time-steps on the MI355X through
libfdtd_hip.so (no XML, no HDF5 dumps, no second process) and ``CalcNF2FF`` evaluates the whole
theta x phi grid from device-accumulated DFT surfaces.  ``compat/openEMS`` and ``compat/CSXCAD`` re-export
this module under the upstream module names, so the reference's solver files run unmodified against
the HIP backend.  Every call is also appended to ``.calls`` in a canonical
vocabulary, which is how tests/ pin this layer against call lists captured from the reference.
"""
from __future__ import annotations

import json
import os
import shutil
import time
from typing import List, Optional, Sequence

import numpy as np

from . import constants
from .grid import RectGrid
from .mesher import mesh_hint_from_box, smooth_mesh_lines, unique_lines
from .scene import Scene, Box as SceneBox, voxelize
from .simulation import Simulation, BoundarySpec
from .nf2ff import calc_nf2ff, NF2FFResult

_AX = {"x": 0, "y": 1, "z": 2, 0: 0, 1: 1, 2: 2}


def _plain(v):
    if isinstance(v, np.ndarray):
        return v.tolist()
    if isinstance(v, (list, tuple)):
        return [_plain(x) for x in v]
    if isinstance(v, np.generic):
        return v.item()
    return v


class _CallLog:
    def __init__(self):
        self.calls: List[dict] = []

    def add(self, op, **kw):
        self.calls.append({"op": op, **{k: _plain(v) for k, v in kw.items()}})


# ---------------------------------------------------------------------------------------------------
# CSXCAD side
# ---------------------------------------------------------------------------------------------------
class CSRectGrid:
    def __init__(self, log: _CallLog):
        self._log = log
        self._unit = 1.0
        self._lines = [[], [], []]
        self.merge_fraction = None      # mesher.smooth_mesh_lines: None = the package default (100), 0 = keep every hint line as openEMS does
        self.merged_lines = [[], [], []]   # per axis: the (line, line) pairs SmoothMeshLines merged into one — where this mesh differs from openEMS's

    def SetDeltaUnit(self, unit):
        self._unit = float(unit)
        self._log.add("SetDeltaUnit", unit=unit)

    def GetDeltaUnit(self):
        return self._unit

    def AddLine(self, ny, lines):
        arr = np.atleast_1d(np.asarray(lines, dtype=float))
        self._lines[_AX[ny]].extend(arr.tolist())
        self._log.add("AddLine", axis="xyz"[_AX[ny]], lines=arr)

    def SetLines(self, ny, lines):
        self._lines[_AX[ny]] = list(np.atleast_1d(np.asarray(lines, dtype=float)))

    def GetLines(self, ny, do_sort=False):
        l = np.asarray(self._lines[_AX[ny]], dtype=float)
        return np.sort(l) if do_sort else l

    def GetQtyLines(self, ny):
        return len(self._lines[_AX[ny]])

    def SmoothMeshLines(self, ny, max_res, ratio=1.5):
        self._log.add("SmoothMeshLines", axis=ny if isinstance(ny, str) else "xyz"[ny], max_res=max_res, ratio=ratio)
        axes = range(3) if ny == "all" else [_AX[ny]]
        for a in axes:
            self._lines[a] = smooth_mesh_lines(self._lines[a], max_res, ratio, merge_fraction=self.merge_fraction,
                                               merged=self.merged_lines[a]).tolist()

    def Sort(self, ny="all"):
        for a in (range(3) if ny == "all" else [_AX[ny]]):
            self._lines[a] = unique_lines(self._lines[a]).tolist()


class CSPrimBox:
    def __init__(self, entry: dict, start, stop, priority):
        self._entry = entry
        self.start = np.asarray(start, dtype=float)
        self.stop = np.asarray(stop, dtype=float)
        self.priority = int(priority)
        self.matrix = np.eye(4)

    def AddTransform(self, kind, *args):
        self._entry.setdefault("transforms", []).append([kind] + [_plain(a) for a in args])
        M = np.eye(4)
        if kind == "Translate":
            M[:3, 3] = np.asarray(args[0], dtype=float)
        elif kind == "RotateAxis":
            ax = _AX[args[0]]
            ang = np.deg2rad(float(args[1]))
            c, s = np.cos(ang), np.sin(ang)
            a1, a2 = (ax + 1) % 3, (ax + 2) % 3
            M[a1, a1] = c; M[a1, a2] = -s; M[a2, a1] = s; M[a2, a2] = c
        else:
            raise ValueError(f"transform '{kind}' is not supported by the HIP backend")
        self.matrix = M @ self.matrix      # applied in the order given, column-vector convention
        return self

    def GetStart(self):
        return self.start

    def GetStop(self):
        return self.stop


class CSProperty:
    def __init__(self, log: _CallLog, kind: str, name: str, **kw):
        self._log, self.kind, self.name = log, kind, name
        self.params = dict(kw)
        self.boxes: List[CSPrimBox] = []
        log.add("Add" + kind, name=name, **kw)

    def GetName(self):
        return self.name

    def AddBox(self, start=None, stop=None, priority=0, **kw):
        self._log.add("AddBox", prop=self.name, priority=priority, start=list(start), stop=list(stop))
        b = CSPrimBox(self._log.calls[-1], start, stop, priority)
        self.boxes.append(b)
        return b


class ContinuousStructure:
    def __init__(self, log: Optional[_CallLog] = None):
        self._log = log or _CallLog()
        self._grid = CSRectGrid(self._log)
        self.properties: List[CSProperty] = []

    def GetGrid(self):
        return self._grid

    def AddMaterial(self, name, **kw):
        p = CSProperty(self._log, "Material", name, **kw)
        self.properties.append(p)
        return p

    def AddMetal(self, name):
        p = CSProperty(self._log, "Metal", name)
        self.properties.append(p)
        return p


# ---------------------------------------------------------------------------------------------------
# openEMS side
# ---------------------------------------------------------------------------------------------------
class UIData:
    def __init__(self, t, val):
        self.ui_time = [np.asarray(t)]
        self.ui_val = [np.asarray(val)]


def dft_time2freq(t, val, freq, signal_type="pulse"):
    """Single-sided spectrum of a sampled pulse ([EXT] openEMS utilities.DFT_time2freq)."""
    t = np.asarray(t, float); val = np.asarray(val, float); freq = np.atleast_1d(np.asarray(freq, float))
    f_val = np.exp(-2j * np.pi * np.outer(freq, t)) @ val
    if signal_type == "pulse":
        f_val = f_val * (t[1] - t[0])
    else:
        f_val = f_val / t.size
    return 2.0 * f_val


def dft_uniform(freq, x, dt, block: int = 128):
    """sum_n x[n] exp(-2 pi j f n dt) for every f, for samples on a uniform time grid, without the len(freq) x len(x) matrix of
    complex exponentials (201 x 12 600 of them were 0.04 s of the 0.29 s plugin call of the reference's default scene):
    n = a * block + b, exp(-j w n dt) = exp(-j w a block dt) * exp(-j w b dt) — two small tables and one real matrix product."""
    freq = np.atleast_1d(np.asarray(freq, float))
    x = np.asarray(x, float)
    n = x.size
    na = (n + block - 1) // block
    xp = np.zeros(na * block)
    xp[:n] = x
    w = -2j * np.pi * freq * dt
    inner = np.exp(np.outer(w, np.arange(block)))            # [f][b]
    outer_ = np.exp(np.outer(w, np.arange(na) * block))       # [f][a]
    part = xp.reshape(na, block) @ inner.T                    # [a][f]
    return np.einsum("fa,af->f", outer_, part)


class LumpedPort:
    """Result side of AddLumpedPort: U/I time series -> incident / reflected waves (row a13 of
    SURVEY §8: the reference's S11 block, microstrip.py:407-426, dead upstream but specified)."""

    def __init__(self, fdtd, number, R, start, stop, exc_dir, excite):
        self._fdtd = fdtd
        self.number, self.R = number, float(R)
        self.start, self.stop = np.asarray(start, float), np.asarray(stop, float)
        self.exc_ny, self.excite = exc_dir, excite
        self.Z_ref = float(R)

    def CalcPort(self, sim_path, freq, ref_impedance=None, ref_plane_shift=None, signal_type="pulse"):
        u, i, dt = self._fdtd._port_series(self.number)
        self.freq = np.atleast_1d(np.asarray(freq, float))
        if ref_impedance is not None:
            self.Z_ref = ref_impedance
        tu = np.arange(u.size) * dt
        ti = (np.arange(i.size) + 0.5) * dt
        self.u_data, self.i_data = UIData(tu, u), UIData(ti, i)
        if u.size == i.size and u.size > 1:
            # uniform time grid: exp(-jw(t + dt/2)) = exp(-jwt) * exp(-jw dt/2), and exp(-jwt) factorised (dft_uniform)
            scale = 2.0 * (dt if signal_type == "pulse" else 1.0 / u.size)
            self.uf_tot = dft_uniform(self.freq, u, dt) * scale
            self.if_tot = dft_uniform(self.freq, i, dt) * np.exp(-1j * np.pi * self.freq * dt) * scale
        else:
            self.uf_tot = dft_time2freq(tu, u, self.freq, signal_type)
            self.if_tot = dft_time2freq(ti, i, self.freq, signal_type)
        self.uf_inc = 0.5 * (self.uf_tot + self.if_tot * self.Z_ref)
        self.if_inc = 0.5 * (self.if_tot + self.uf_tot / self.Z_ref)
        self.uf_ref = self.uf_tot - self.uf_inc
        self.if_ref = self.if_inc - self.if_tot
        self.P_inc = 0.5 * np.real(self.uf_inc * np.conj(self.if_inc))
        self.P_ref = 0.5 * np.real(self.uf_ref * np.conj(self.if_ref))
        self.P_acc = 0.5 * np.real(self.uf_tot * np.conj(self.if_tot))
        return self


class nf2ff:
    def __init__(self, fdtd, name="nf2ff", start=None, stop=None):
        self._fdtd, self.name = fdtd, name
        self.start, self.stop = start, stop

    def can_evaluate(self, freq) -> bool:
        """Whether CalcNF2FF can answer for `freq` after this run: any frequency inside the recorder's band when the faces
        were recorded in the time domain (the default) or when the running DFT took the comb it snaps to; else only the
        frequencies the running DFT accumulated."""
        sim = self._fdtd.sim
        if sim is None or sim.nf2ff_box is None:
            return False
        f = np.atleast_1d(np.asarray(freq, float))
        if sim.nf2ff_mode == "record":
            return bool(np.all(f <= sim.nf2ff_fmax * (1 + 1e-9)))
        if self._fdtd._nf2ff_snap:
            return True
        rec = np.asarray(sim.nf2ff_freqs, float)
        return bool(all(np.min(np.abs(rec - x)) <= 1e-6 * max(x, 1.0) for x in f))

    def CalcNF2FF(self, sim_path, freq, theta, phi, radius=1, center=(0, 0, 0), outfile=None,
                  read_cached=False, verbose=0):
        """theta/phi in DEGREES, center in metres (fixed.py:296).  Returns an object with the upstream
        attribute names; arrays are [frequency] lists of (ntheta, nphi).
        Multi-rank runs: Run() has already reduced the faces over the ranks for the excitation's centre frequency (and
        nf2ff_freqs), so a call for those is LOCAL and may be made on rank 0 alone, as upstream post-processing is.  Any
        OTHER frequency (time-domain recording) has to be transformed on every rank's slab first: such a call is a
        collective — every rank must make it, with the same frequencies."""
        return self._fdtd._calc_nf2ff(np.atleast_1d(np.asarray(freq, float)), np.atleast_1d(np.asarray(theta, float)),
                                      np.atleast_1d(np.asarray(phi, float)), float(radius),
                                      np.asarray(center, float))


def nf2ff_comb(f0: float, every: int = 10) -> np.ndarray:
    """Every `every`-th point of the reference's S11 grid linspace(max(1 GHz, 0.7 f0), 1.3 f0, 201)
    (solver_fdtd_openems_microstrip.py:408) plus f0 itself: what the NF2FF faces accumulate when the run is too
    long for time-domain recording and the caller named no frequencies."""
    f = np.linspace(max(1e9, 0.7 * f0), 1.3 * f0, 201)[::every]
    return np.unique(np.append(f, f0))


class openEMS:
    """FDTD front object.  Backend options beyond the upstream signature are keyword-only:
    n_gpus / rank / world (z-slab decomposition), device, cpml_cells (default: the _N of 'PML_N'),
    nf2ff_mode: 'auto' (default) records the NF2FF faces in the time domain in HBM when that fits the budget, so
    that CalcNF2FF can be asked for ANY frequency after the run, as upstream; otherwise / 'dft': running DFT at
    nf2ff_freqs (default then: a 21-point comb over the reference's S11 band, CalcNF2FF snaps to the nearest)."""

    def __init__(self, NrTS=1e9, EndCriteria=1e-5, *, lib=None, device=0, rank=0, world=1, cpml_cells=None,
                 nf2ff_freqs=None, nf2ff_mode="auto", comm=None, **kw):
        self.calls_log = _CallLog()
        self.NrTS, self.EndCriteria = NrTS, EndCriteria
        self.calls_log.add("openEMS", NrTS=NrTS, EndCriteria=EndCriteria)
        self._lib, self._device, self._rank, self._world = lib, device, rank, world
        self._cpml_cells, self._nf2ff_freqs, self._comm = cpml_cells, nf2ff_freqs, comm
        self._nf2ff_mode, self._nf2ff_snap, self._box_cache = nf2ff_mode, False, {}
        self._csx: Optional[ContinuousStructure] = None
        self._bc = ["PEC"] * 6
        self._f0 = self._fc = None
        self._edge_hints = []
        self._ports: List[LumpedPort] = []
        self._nf2ff: Optional[nf2ff] = None
        self.sim: Optional[Simulation] = None
        self.stats = None

    @property
    def calls(self):
        return self.calls_log.calls

    # -- set-up ---------------------------------------------------------------------------------------
    def SetGaussExcite(self, f0, fc):
        self._f0, self._fc = float(f0), float(fc)
        self.calls_log.add("SetGaussExcite", f0=f0, fc=fc)

    def SetBoundaryCond(self, BC):
        self._bc = list(BC)
        self.calls_log.add("SetBoundaryCond", bc=list(BC))

    def SetCSX(self, CSX: ContinuousStructure):
        # one shared call log, in call order
        self.calls_log.calls.extend(CSX._log.calls)
        CSX._log = self.calls_log
        CSX._grid._log = self.calls_log
        for p in CSX.properties:
            p._log = self.calls_log
        self._csx = CSX

    def GetCSX(self):
        return self._csx

    def AddEdges2Grid(self, dirs, primitives=None, properties=None, **kw):
        mer = kw.get("metal_edge_res")
        self.calls_log.add("AddEdges2Grid", dirs=dirs, prop=getattr(properties, "name", None), metal_edge_res=mer)
        d = [_AX[c] for c in dirs] if isinstance(dirs, str) and dirs != "all" else ([0, 1, 2] if dirs == "all" else [_AX[c] for c in dirs])
        boxes = []
        if properties is not None:
            for p in (properties if isinstance(properties, (list, tuple)) else [properties]):
                boxes.extend(p.boxes)
        if primitives is not None:
            boxes.extend(primitives if isinstance(primitives, (list, tuple)) else [primitives])
        grid = self._csx.GetGrid()
        for b in boxes:
            if not np.allclose(b.matrix, np.eye(4)):
                continue   # as upstream: edge hints cannot be derived for transformed primitives
            hint = mesh_hint_from_box(b.start, b.stop, d, mer)
            for a in range(3):
                if hint[a]:
                    grid._lines[a].extend(hint[a])

    def AddLumpedPort(self, port_nr, R, start, stop, p_dir, excite=0, **kw):
        self.calls_log.add("AddLumpedPort", port_nr=port_nr, R=R, start=list(start), stop=list(stop), p_dir=p_dir,
                           excite=excite, priority=kw.get("priority", 0), edges2grid=kw.get("edges2grid"))
        port = LumpedPort(self, port_nr, R, start, stop, _AX[p_dir], excite)
        port.priority = kw.get("priority", 0)
        self._ports.append(port)
        e2g = kw.get("edges2grid")
        if e2g:
            dirs = [0, 1, 2] if e2g == "all" else [_AX[c] for c in e2g]
            hint = mesh_hint_from_box(start, stop, dirs, None)
            for a in range(3):
                if hint[a]:
                    self._csx.GetGrid()._lines[a].extend(hint[a])
        return port

    def CreateNF2FFBox(self, name="nf2ff", start=None, stop=None, **kw):
        self.calls_log.add("CreateNF2FFBox")
        self._nf2ff = nf2ff(self, name, start, stop)
        return self._nf2ff

    # -- run ------------------------------------------------------------------------------------------
    def _build_scene(self):
        csx = self._csx
        unit = csx.GetGrid().GetDeltaUnit()
        lines = [unique_lines(csx.GetGrid()._lines[a]) * unit for a in range(3)]
        grid = RectGrid(*lines)
        sc = Scene(unit=unit)
        for p in csx.properties:
            if p.kind == "Material":
                m = sc.add_material(p.name, p.params.get("epsilon", 1.0), p.params.get("kappa", 0.0))
                for b in p.boxes:
                    m.boxes.append(SceneBox(tuple(b.start), tuple(b.stop), b.priority, b.matrix.copy()))
            else:
                m = sc.add_metal(p.name)
                for b in p.boxes:
                    m.boxes.append(SceneBox(tuple(b.start), tuple(b.stop), b.priority, b.matrix.copy()))
        for port in self._ports:
            sc.add_lumped_port(port.number, port.R, port.start, port.stop, port.exc_ny, port.excite, port.priority)
        return grid, sc

    @staticmethod
    def _wipe_sim_path(sim_path):
        """cleanup=True, as upstream's engine does before a run (fixed.py:280): a pre-existing sim_path goes.  Never a
        directory this process runs in (or any of its parents), never a filesystem root."""
        path = os.path.realpath(sim_path)
        cwd = os.path.realpath(os.getcwd())
        if not os.path.isdir(path) or path == os.path.dirname(path) or cwd == path or cwd.startswith(path + os.sep):
            return
        shutil.rmtree(path, ignore_errors=True)

    def Run(self, sim_path, cleanup=False, setup_only=False, verbose=None, **kw):
        """Time-step on the GPU.  Blocks; ctypes releases the GIL so a GUI thread stays live
        (the reference calls this from one background thread, gui_app.py:2688-2690)."""
        from ._capi import load_hip_library
        self.calls_log.add("Run", verbose=verbose, cleanup=cleanup)
        if self._csx is None or self._f0 is None:
            raise RuntimeError("SetCSX and SetGaussExcite must be called before Run")
        if cleanup and sim_path and self._rank == 0:
            self._wipe_sim_path(sim_path)
        lib = self._lib or load_hip_library()
        grid, sc = self._build_scene()
        vox = voxelize(sc, grid)
        bc = BoundarySpec.parse(self._bc, self._cpml_cells)
        freqs = None
        if self._nf2ff is not None:
            freqs = [self._f0] if self._nf2ff_freqs is None else list(self._nf2ff_freqs)

        def make(fr):
            return Simulation(grid, vox, f0=self._f0, fc=self._fc, boundary=bc, nr_ts=int(min(self.NrTS, 2**31 - 2)),
                              end_criteria=float(self.EndCriteria), nf2ff_freqs=fr, nf2ff_mode=self._nf2ff_mode)
        self.sim = make(freqs)
        self._nf2ff_snap, self._box_cache = False, {}
        if self._nf2ff is not None and self.sim.nf2ff_mode == "dft" and self._nf2ff_freqs is None:
            self.sim = make(nf2ff_comb(self._f0))     # no time-domain record: a comb, CalcNF2FF snaps to it
            self._nf2ff_snap = True
        self.sim.build(lib, rank=self._rank, world=self._world, device=self._device)
        if self._comm is not None:
            self._comm.attach(self.sim)
        if setup_only:
            return
        allreduce = self._comm.allreduce if self._comm is not None else None
        self.stats = self.sim.run(verbose=int(verbose or 0), allreduce=allreduce)
        self._u_i = self.sim.port_series(allreduce)
        self._allreduce = allreduce
        self._boxes = None
        if self._nf2ff is not None and self.sim.nf2ff_mode == "dft":
            self._boxes = self.sim.nf2ff_boxes(allreduce)
        elif self._nf2ff is not None and allreduce is not None:
            # several ranks, faces recorded in the time domain: transform and reduce the frequencies the caller is known to
            # ask for HERE, where every rank is, so that CalcNF2FF at those is local (rank 0 alone may post-process)
            f = np.asarray(freqs, float)
            self._box_cache = {tuple(float(x) for x in f): self.sim.nf2ff_boxes(allreduce, freqs=f)}
        if sim_path and self._rank == 0:
            try:
                os.makedirs(sim_path, exist_ok=True)
                nx, ny, nz = grid.shape
                with open(os.path.join(sim_path, "fdtd_hip_run.json"), "w") as fh:
                    json.dump({"grid": [nx, ny, nz], "cells": grid.ncells, "dt": self.sim.dt,
                               "steps": self.stats.steps, "seconds": self.stats.seconds,
                               "mcells_per_s": self.stats.mcells_per_s, "energy_db": float(self.stats.energy_db),
                               "operator": self.sim.operator_form, "n_gpus": self._world,
                               "halo_transport": getattr(self._comm, "transport_used", None),
                               "halo_transports_failed": list(self.stats.transports_failed),
                               "schedule_fallback": self.stats.schedule_fallback}, fh)
                # the port series as upstream's probe files (text): only on request — formatting them takes 0.03 s of a
                # 0.34 s call on the reference's default scene, and CalcPort reads the arrays, not the files
                if int(verbose or 0) > 0 or os.environ.get("FDTD_WRITE_PORT_FILES"):
                    for port, (u, i) in zip(self._ports, self._u_i):
                        np.savetxt(os.path.join(sim_path, f"port_ut{port.number}"), np.c_[np.arange(u.size) * self.sim.dt, u])
                        np.savetxt(os.path.join(sim_path, f"port_it{port.number}"), np.c_[(np.arange(i.size) + 0.5) * self.sim.dt, i])
            except OSError:
                pass

    # -- results --------------------------------------------------------------------------------------
    def _port_series(self, number):
        if self.sim is None:
            raise RuntimeError("Run() first")
        for port, (u, i) in zip(self._ports, self._u_i):
            if port.number == number:
                return u, i, self.sim.dt
        raise KeyError(number)

    def _calc_nf2ff(self, freq, theta_deg, phi_deg, radius, center):
        if self.sim is None or self._nf2ff is None or self.sim.nf2ff_box is None:
            raise RuntimeError("Run() with an NF2FF box first")
        lib = self.sim.lib
        if self.sim.nf2ff_mode == "record":
            # time-domain faces in HBM: transform them for exactly the frequencies asked for (as upstream's nf2ff does
            # with the dumps); cached, since the 3-D variants call once per phi (microstrip_3d.py:224-225)
            key = tuple(float(f) for f in freq)
            if key not in self._box_cache:
                hit = [k for k in self._box_cache if all(any(abs(f - g) <= 1e-9 * max(abs(g), 1.0) for g in k) for f in key)]
                if hit:      # a subset of what is cached (e.g. Run's reduced set): pick the rows
                    src = hit[0]
                    rows = [int(np.argmin(np.abs(np.asarray(src) - f))) for f in key]
                    self._box_cache[key] = [b[rows] for b in self._box_cache[src]]
                else:        # (with several ranks this is a collective: see CalcNF2FF)
                    self._box_cache = {key: self.sim.nf2ff_boxes(self._allreduce, freqs=freq)}
            boxes, f_used = self._box_cache[key], freq
        else:
            rec = self.sim.nf2ff_freqs
            idx = []
            for f in freq:
                k = int(np.argmin(np.abs(rec - f)))
                if abs(rec[k] - f) > 1e-6 * max(f, 1.0) and not self._nf2ff_snap:
                    raise ValueError(f"NF2FF frequency {f:g} Hz was not recorded (recorded: {rec.tolist()}); "
                                     "pass nf2ff_freqs=[...] or nf2ff_mode='record' to openEMS(...)")
                idx.append(k)
            boxes, f_used = [b[idx] for b in self._boxes], rec[idx]
        res = calc_nf2ff(lib, self.sim.nf2ff_box, boxes, f_used, np.deg2rad(theta_deg),
                         np.deg2rad(phi_deg), center, device=self._device)
        out = _NF2FFResults()
        out.theta, out.phi, out.r, out.freq = res.theta, res.phi, radius, res.freq
        out.Dmax = np.asarray(res.Dmax)
        out.Prad = np.asarray(res.Prad)
        out.E_theta = [e / radius for e in res.E_theta]
        out.E_phi = [e / radius for e in res.E_phi]
        out.E_norm = [e / radius for e in res.E_norm]
        out.E_cprh = [(et + 1j * ep) / np.sqrt(2.0) / radius for et, ep in zip(res.E_theta, res.E_phi)]
        out.E_cplh = [(et - 1j * ep) / np.sqrt(2.0) / radius for et, ep in zip(res.E_theta, res.E_phi)]
        out.P_rad = [u / radius ** 2 for u in res.P_rad]
        return out


class _NF2FFResults:
    pass


# upstream module layout: openEMS.openEMS, openEMS.physical_constants, CSXCAD.ContinuousStructure
class physical_constants:
    C0 = constants.C0
    MUE0 = constants.MU0
    EPS0 = constants.EPS0
    Z0 = constants.ETA0

"""Assembly of one FDTD run: operator + boundaries + ports + recording surfaces -> engine(s).

This is the host-side counterpart of what happens between the reference's ``prepare_*`` and the
end of ``FDTD.Run(...)`` (antenna_sim/solver_fdtd_openems_fixed.py:171-220,280): everything the
external engine derives from the scene before and while stepping.  The compute itself is
libfdtd_hip.so (``lib`` argument = a library exporting include/fdtd_hip.h).
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import List, Optional, Sequence
import numpy as np

from .constants import C0, EPS0, MU0
from .grid import RectGrid
from .scene import VoxelScene
from .ecoperator import build_operator, ECOperator, metric_lists, pack_metric_tables, lumped_overrides
from .cpml import CPMLSpec, build_cpml
from .excitation import gauss_pulse, dft_twiddles
from .nf2ff import NF2FFBox
from . import _capi
from ._capi import Engine, KIND_V, KIND_I


def slab_range(nz: int, world: int, rank: int):
    """Contiguous z-planes [k0, k0+nk) of `rank`, planes counted alike; the remainder goes to the first ranks."""
    base, rem = divmod(nz, world)
    k0 = rank * base + min(rank, rem)
    return k0, base + (1 if rank < rem else 0)


# A z-plane inside a z-directed CPML layer reads and writes four more psi arrays per timestep (+16 of 37 bytes per cell and
# half-step) than a plane outside: measured on MI355X it costs 1.3-1.55x as much (NS: 20 layer planes +10.4 us over 60 planes of
# 0.95 us; C3: +18.0 us over 80 planes of 1.78 us — profiles/r02/cpml_axis_cost.txt; per-slab times: profiles/r03/slab_balance.txt).
# With planes counted alike the first and the last rank, which own ALL layer planes, take up to 37 % longer than the interior
# ranks (C5 on 8 GPUs: 10 of their 15 planes), and every rank waits for its neighbours' halos each half-step.
# (weight: each slab timed alone, profiles/r03/slab_balance*.txt — 1.4 leaves max / mean at 1.03 (NS over 8), 1.06 (C4 over 4), 1.03 (C5 over 8);
# 1.5 is better for C4 (1.045) and worse for thin slabs, whose fixed per-launch cost the per-plane model does not know: NS over 4 1.07)
Z_LAYER_PLANE_COST = float(os.environ.get("FDTD_SLAB_WZ", "1.4"))


def plane_costs(nz: int, cpml_lo: int = 0, cpml_hi: int = 0, w_layer: Optional[float] = None) -> np.ndarray:
    """Relative cost of every z-plane of the global grid: 1 outside the z-directed CPML layers, `w_layer` inside
    (planes 0 .. cpml_lo-1 and nz-1-cpml_hi .. nz-1, the planes build_cpml gives psi slots)."""
    w = np.ones(nz)
    wl = Z_LAYER_PLANE_COST if w_layer is None else float(w_layer)
    if cpml_lo:
        w[:cpml_lo] = wl
    if cpml_hi:
        w[nz - 1 - cpml_hi:] = wl
    return w


def slab_partition(costs: Sequence[float], world: int, min_planes: int = 2):
    """[(k0, nk)] * world: contiguous slabs of at least `min_planes` planes whose largest summed cost is as small as possible
    (exact: dynamic programme over the cut positions; ties go to the partition with the smaller sum of squared costs).
    Deterministic, so every rank computes the same cuts."""
    c = np.asarray(costs, float)
    nz = c.size
    if world < 1 or nz < world * min_planes:
        raise ValueError(f"{nz} planes cannot be cut into {world} slabs of >= {min_planes} planes")
    pre = np.concatenate([[0.0], np.cumsum(c)])
    INF = float("inf")
    # best[r][k] = (largest slab cost, sum of squares) of the first k planes in r slabs
    best = [[(INF, INF)] * (nz + 1) for _ in range(world + 1)]
    cut = [[-1] * (nz + 1) for _ in range(world + 1)]
    best[0][0] = (0.0, 0.0)
    for r in range(1, world + 1):
        for k in range(r * min_planes, nz - (world - r) * min_planes + 1):
            for q in range((r - 1) * min_planes, k - min_planes + 1):
                m, sq = best[r - 1][q]
                if m == INF:
                    continue
                sc = pre[k] - pre[q]
                cand = (max(m, sc), sq + sc * sc)
                if cand[0] < best[r][k][0] - 1e-12 or (abs(cand[0] - best[r][k][0]) <= 1e-12 and cand[1] < best[r][k][1] - 1e-12):
                    best[r][k], cut[r][k] = cand, q
    out, k = [], nz
    for r in range(world, 0, -1):
        q = cut[r][k]
        out.append((q, k - q))
        k = q
    return out[::-1]


@dataclass
class BoundarySpec:
    """Per face (x-,x+,y-,y+,z-,z+): 'PEC', 'MUR' or 'CPML' (reference strings 'MUR' / 'PML_8',
    solver_fdtd_openems_microstrip_3d.py:84)."""
    kinds: Sequence[str] = ("CPML",) * 6
    cpml_cells: int = 10
    cpml: CPMLSpec = field(default_factory=CPMLSpec)

    @classmethod
    def parse(cls, boundary, cpml_cells: Optional[int] = None) -> "BoundarySpec":
        if isinstance(boundary, BoundarySpec):
            return boundary
        names = [boundary] * 6 if isinstance(boundary, (str, int)) else list(boundary)
        kinds, cells = [], cpml_cells
        for b in names:
            s = str(b).upper()
            if s.startswith("MUR") or s == "2":
                kinds.append("MUR")
            elif s.startswith("PML") or s.startswith("CPML") or s == "3":
                kinds.append("CPML")
                if cells is None and "_" in s and s.split("_")[1].isdigit():
                    cells = int(s.split("_")[1])
            elif s in ("PEC", "0"):
                kinds.append("PEC")
            else:
                raise ValueError(f"unsupported boundary '{b}'")
        return cls(tuple(kinds), 8 if cells is None else cells)

    def face_cells(self):
        return tuple(self.cpml_cells if k == "CPML" else 0 for k in self.kinds)


@dataclass
class RunStats:
    steps: int = 0
    seconds: float = 0.0
    mcells_per_s: float = 0.0
    energy_db: float = 0.0
    stopped_by_energy: bool = False
    schedule_fallback: Optional[str] = None   # set when the run was repeated under the two-launch schedule (see Simulation.run)
    transports_failed: tuple = ()             # decomposed run: halo transports that set up but failed in the first timesteps (see Simulation.run)
    transport_failure_reasons: tuple = ()     # ... and what the library said each time


class Simulation:
    def __init__(self, grid: RectGrid, vox: VoxelScene, *, f0: float, fc: float, boundary="CPML",
                 cpml_cells: Optional[int] = None, nr_ts: int = 30000, end_criteria: float = 1e-4,
                 dt: Optional[float] = None, nf2ff_freqs: Optional[Sequence[float]] = None,
                 nf2ff_inset: Optional[int] = None, dft_oversample: float = 4.0, use_classes: bool = True,
                 device_operator: bool = True, nf2ff_mode: str = "dft", rec_budget_bytes: Optional[int] = None):
        self.grid, self.vox = grid, vox
        self.f0, self.fc = float(f0), float(fc)
        self.bc = BoundarySpec.parse(boundary, cpml_cells)
        self.nr_ts, self.end_criteria = int(nr_ts), float(end_criteria)
        self.dt = grid.courant_dt() if dt is None else float(dt)
        self.use_classes = use_classes
        # True: the engine builds the operator itself from materials + mesh (fdtd_build_operator: on the GPU for
        # libfdtd_hip.so); False: numpy build on the host (ecoperator.build_operator, the spec) + array upload
        self.device_operator = device_operator
        self._op: Optional[ECOperator] = None
        cells = self.bc.face_cells()
        self.cpml = None
        if any(cells):
            spec = CPMLSpec(**{**self.bc.cpml.__dict__, "cells": cells})
            self.cpml = build_cpml(grid, self.dt, spec)
        self.mur_enable = np.array([1 if k == "MUR" else 0 for k in self.bc.kinds], np.int32)
        self.mur_coeff_f64 = np.zeros(6, np.float64)
        for f in range(6):
            l = grid.lines[f // 2]
            delta = (l[-1] - l[-2]) if f % 2 else (l[1] - l[0])
            self.mur_coeff_f64[f] = (C0 * self.dt - delta) / (C0 * self.dt + delta)
        self.mur_coeff = self.mur_coeff_f64.astype(np.float32)   # what the C ABI takes
        self.signal = gauss_pulse(self.f0, self.fc, self.dt)
        # (openEMS prints "Requested excitation pulse would be N timesteps ... Cutting to max number of timesteps!" here and goes on; so do we)
        self.excitation_warning = None
        if len(self.signal) > self.nr_ts:
            self.excitation_warning = (f"the excitation pulse is {len(self.signal)} timesteps long (dt = {self.dt:.3e} s: smallest cell "
                                       f"{min(float(np.min(np.diff(l))) for l in grid.lines) * 1e6:.1f} um) but NrTS = {self.nr_ts}: the run ends before the pulse does")
            import warnings
            warnings.warn(self.excitation_warning, RuntimeWarning, stacklevel=2)
        # NF2FF recording
        self.nf2ff_box: Optional[NF2FFBox] = None
        self.nf2ff_warning: Optional[str] = None
        self.nf2ff_freqs = None
        self.nf2ff_mode, self.rec_bytes, self.nf2ff_fmax = "dft", 0, 0.0
        if nf2ff_freqs is not None:
            self.nf2ff_freqs = np.atleast_1d(np.asarray(nf2ff_freqs, float))
            n = grid.shape
            lo, hi = [], []
            for a in range(3):
                il = (cells[2 * a] + 2) if nf2ff_inset is None else nf2ff_inset
                ih = (cells[2 * a + 1] + 2) if nf2ff_inset is None else nf2ff_inset
                il = max(il, 3); ih = max(ih, 3)
                lo.append(il); hi.append(n[a] - 1 - ih)
            self.nf2ff_box = NF2FFBox(grid, lo, hi)
            # a Huygens surface that runs ALONG metal is no closed surface around the sources: the far field computed from it is not
            # the antenna's (a ground plane on the very plane of the lower face made a 6 dBi patch read 14.6 dBi).  Metal that merely
            # CROSSES a face (an infinite ground plane, as in the legacy scene) is the caller's modelling decision and not flagged.
            self.nf2ff_warning = None
            for a in range(3):
                for side, pos in ((0, lo[a]), (1, hi[a])):
                    sl = [slice(lo[2], hi[2] + 1), slice(lo[1], hi[1] + 1), slice(lo[0], hi[0] + 1)]
                    sl[2 - a] = pos
                    frac = max(float(vox.pec[t][tuple(sl)].mean()) for t in range(3) if t != a)
                    if frac > 0.05 and self.nf2ff_warning is None:
                        self.nf2ff_warning = (f"{100 * frac:.0f} % of the NF2FF box's {'xyz'[a]}-{'max' if side else 'min'} face (node plane {pos}) "
                                              f"lies on metal edges: the far field will not be the antenna's; move the structure or the box")
                        import warnings
                        warnings.warn(self.nf2ff_warning, RuntimeWarning, stacklevel=2)
            fmax = max(self.f0 + self.fc, float(np.max(self.nf2ff_freqs)))
            self.dft_every = max(1, int(np.floor(1.0 / (2.0 * fmax * dft_oversample * self.dt))))
            self.dft_nsamples = self.nr_ts // self.dft_every + 1
            # "record": the faces keep float32 time-domain samples in HBM and any frequency <= fmax can be asked for
            # after the run (what CalcNF2FF does with the engine's dumps); "dft": running sums at nf2ff_freqs only
            # (constant memory).  "auto" records when the samples fit the budget (default 32 GiB of 288 GB per GPU).
            if nf2ff_mode not in ("dft", "record", "auto"):
                raise ValueError("nf2ff_mode must be 'dft', 'record' or 'auto'")
            self.rec_bytes = 4 * self.dft_nsamples * sum(
                int(np.prod([r.hi[a] - r.lo[a] + 1 for a in range(3)])) for r in self.nf2ff_box.requests)
            budget = int(os.environ.get("FDTD_REC_BUDGET_BYTES", 32 << 30)) if rec_budget_bytes is None else int(rec_budget_bytes)
            self.nf2ff_mode = nf2ff_mode if nf2ff_mode != "auto" else ("record" if self.rec_bytes <= budget else "dft")
            self.nf2ff_fmax = fmax
        self.engine: Optional[Engine] = None
        self.lib = None
        self.external_transport = None     # distributed.SlabComm when halos travel through the host
        self._port_probe_ids = []
        self._nf_ids = []

    @property
    def op(self) -> ECOperator:
        """The operator in its host (numpy) formulation — built on first use; the default product path never asks."""
        if self._op is None:
            v = self.vox
            self._op = build_operator(self.grid, v.eps_r, v.kappa, v.pec, self.dt, v.lumped)
        return self._op

    # ---------------------------------------------------------------------------------------------
    def slabs(self, world: int, partition: str = "cost"):
        """[(k0, nk)] of every rank.  "cost": slabs of equal COST — the z-layer planes weigh Z_LAYER_PLANE_COST, so the two
        end ranks own fewer planes; "even": equal plane counts (SURVEY §8e's first cut)."""
        nz = self.grid.shape[2]
        cells = self.bc.face_cells()
        if partition == "even" or world == 1 or not (cells[4] or cells[5]):
            return [slab_range(nz, world, r) for r in range(world)]
        if partition != "cost":
            raise ValueError("partition must be 'cost' or 'even'")
        return slab_partition(plane_costs(nz, cells[4], cells[5]), world)

    def build(self, lib, *, rank: int = 0, world: int = 1, device: int = 0, flags: int = 0, partition: str = "cost") -> Engine:
        g = self.grid
        nx, ny, nz = g.shape
        k0, nk = self.slabs(world, partition)[rank]
        self.partition = partition
        if nk < 2:
            raise ValueError(f"slab of rank {rank} has {nk} planes; need >= 2")
        e = Engine(lib, nx, ny, nz, self.dt, k0=k0, nk=nk, rank=rank, world=world, device=device,
                   max_steps=self.nr_ts, flags=flags)
        if self.device_operator:
            v = self.vox
            emet, hmet = pack_metric_tables(*metric_lists(g, self.dt), g, k0, nk)
            e.build_operator(g.d, v.eps_r, v.kappa, v.pec, EPS0,
                             lumped_overrides(g, v.eps_r, v.kappa, v.pec, self.dt, v.lumped), emet, hmet,
                             prefer_classes=self.use_classes)
            self.operator_form = "raw" if e.operator_form()[0] == "raw" else "classes"
        else:
            cls = self.op.classes(k0, nk) if self.use_classes else None
            if cls is not None:
                emet, hmet = self.op.metric_tables(k0, nk)
                e.set_operator_classes(cls[0], cls[1], cls[2], emet, hmet)
                self.operator_form = "classes"
            else:
                e.set_operator_raw(*self.op.raw(k0, nk))
                self.operator_form = "raw"
        if self.cpml is not None:
            e.set_cpml(*self.cpml.for_slab(k0, nk))
        if self.mur_enable.any():
            e.set_mur(self.mur_enable, self.mur_coeff)
        e.set_signal(self.signal)
        self._port_probe_ids = []
        for p in self.vox.ports:
            if p.port.excite != 0:
                e.add_source(p.src_idx, p.src_comp, p.src_amp,
                             np.full(p.src_idx.size, p.port.delay_steps, np.int32))
            uid = e.add_probe(KIND_V, p.v_idx, p.v_comp, p.v_w)
            iid = e.add_probe(KIND_I, p.i_idx, p.i_comp, p.i_w)
            self._port_probe_ids.append((uid, iid))
        if self.nf2ff_box is not None:
            if self.nf2ff_mode == "record":
                try:
                    e.set_recorder(self.dft_every, self.dft_nsamples)
                    self._nf_ids = self.nf2ff_box.register(e)
                except _capi.FdtdError as err:
                    raise _capi.FdtdError(f"{err} — time-domain NF2FF recording needs {self.rec_bytes / 2**30:.1f} GiB; "
                                          "use nf2ff_mode='dft'") from err
            else:
                tw_v = dft_twiddles(self.nf2ff_freqs, self.dt, self.dft_every, self.dft_nsamples, 0.0)
                tw_i = dft_twiddles(self.nf2ff_freqs, self.dt, self.dft_every, self.dft_nsamples, 0.5)
                e.set_dft(self.dft_every, tw_v, tw_i)
                self._nf_ids = self.nf2ff_box.register(e)
        self.engine, self.lib = e, lib
        self.rank, self.world, self.device = rank, world, device
        self._build_flags = int(flags)
        return e

    # ---------------------------------------------------------------------------------------------
    def run(self, *, max_steps: Optional[int] = None, check_every: int = 200, verbose: int = 0,
            allreduce=None, log=print) -> RunStats:
        """Step until nr_ts or until the field energy drops below end_criteria x its maximum
        ([EXT] openEMS end criterion, EndCriteria=1e-4 at solver_fdtd_openems_fixed.py:171).
        `allreduce(np.ndarray) -> np.ndarray` sums the two energy terms over ranks when world > 1."""
        import time
        e = self.engine
        total = self.nr_ts if max_steps is None else min(max_steps, self.nr_ts)
        emax, stats = 0.0, RunStats()
        t0 = time.perf_counter()
        done = e.step
        fresh = done == 0          # stepping from the state build() left: a repeat from scratch reproduces it
        comm = getattr(self, "comm", None)      # distributed.SlabComm.attach leaves itself here
        while done < total:
            n = min(check_every, total - done)
            if self.external_transport is not None:
                self.external_transport.run_steps(e, n)
            elif self.world > 1 and fresh and done == 0 and comm is not None and comm.transport == "auto":
                # Decomposed run, first timesteps: the probe of the halo transport the ranks agreed on.  One that set up (and passed
                # its self-test) but errors once timesteps depend on it — every halo wait is bounded — sends ALL ranks to the next
                # transport (p2p -> rccl -> host): new contexts, from the initial state.
                # What failed decides what happens, and every rank does the same (one all-reduce of a code):
                #   1  a SCHEDULE error (a flag wait of the one-launch schedule ran out): same transport, two launches per timestep;
                #   2  a TRANSPORT error (a halo wait ran out, the transport is refused on this topology, an RCCL error): the next transport;
                #   3  anything else (out of memory, a bad argument, a device fault): raised, on every rank — not papered over.
                code, why = 0, ""
                try:
                    e.run(n)
                except _capi.FdtdError as exc:
                    why = str(exc)
                    low = why.lower()
                    code = 1 if "wavefront schedule" in low else 2 if ("p2p" in low or "nccl" in low or "rccl" in low) else 3
                worst = int(round(float(np.max(comm.allreduce_max(np.array([float(code)]))))))
                if worst == 3:
                    raise _capi.FdtdError(why or f"rank {self.rank}: another rank's engine failed in the first timesteps of the decomposed run")
                if worst:
                    log(f"[fdtd-hip rank {self.rank}] {'one-launch schedule' if worst == 1 else 'halo transport ' + str(comm.transport_used)} failed at run time"
                        + (f" ({why})" if why else "") + (" — every rank repeats under two launches per timestep" if worst == 1 else " — every rank takes the next one"))
                    if worst == 1:
                        stats.schedule_fallback = why or "another rank's one-launch schedule timed out"
                        self._build_flags = (self._build_flags & ~_capi.FLAG_KERNEL_MASK) | _capi.FLAG_KERNEL_DIRECT
                    else:
                        stats.transports_failed += (comm.transport_used,)
                        stats.transport_failure_reasons += (why or "failed on another rank",)
                        comm.skip.add(comm.transport_used)
                    comm.barrier()                  # nobody frees a mailbox a neighbour may still write into
                    e.close()
                    e = self.build(self.lib, rank=self.rank, world=self.world, device=self.device, partition=self.partition,
                                   flags=self._build_flags)
                    comm.attach(self)
                    emax = 0.0
                    continue
            else:
                try:
                    e.run(n)
                except _capi.FdtdError as exc:
                    # The one-launch-per-timestep schedule depends on workgroups being dispatched in order; a block that
                    # waits too long for an earlier block's flag sets an error word and the run comes back invalid
                    # (never a hang).  Heal it here: a new context under the two-launch schedule (no flags, no
                    # dependence on dispatch order), same process, from the initial state — once.
                    if not (fresh and self.world == 1 and ("wavefront schedule" in str(exc) or "resident schedule" in str(exc))
                            and stats.schedule_fallback is None):
                        raise
                    log(f"[fdtd-hip] {exc} — repeating the run under the two-launch schedule")
                    e.close()
                    e = self.build(self.lib, rank=self.rank, world=self.world, device=self.device, partition=self.partition,
                                   flags=(self._build_flags & ~_capi.FLAG_KERNEL_MASK) | _capi.FLAG_KERNEL_DIRECT)
                    stats.schedule_fallback = str(exc)
                    done, emax = 0, 0.0
                    continue
            done += n
            sv, si = e.energy()
            s = np.array([sv, si])
            if allreduce is not None:
                s = allreduce(s)
            en = EPS0 * s[0] + MU0 * s[1]
            emax = max(emax, en)
            ratio = en / emax if emax > 0 else 1.0
            stats.energy_db = 10.0 * np.log10(max(ratio, 1e-300))
            if verbose:
                el = time.perf_counter() - t0
                log(f"[fdtd-hip] step {done:6d}/{total}  energy {stats.energy_db:7.2f} dB  "
                    f"{self.grid.ncells * done / max(el, 1e-9) / 1e6:9.1f} MC/s")
            if self.end_criteria > 0 and done >= len(self.signal) and ratio < self.end_criteria:
                stats.stopped_by_energy = True
                break
        stats.steps = done
        stats.seconds = time.perf_counter() - t0
        stats.mcells_per_s = self.grid.ncells * done / max(stats.seconds, 1e-9) / 1e6
        return stats

    # ---------------------------------------------------------------------------------------------
    def port_series(self, allreduce=None):
        """[(u(t), i(t))] per port; i is sampled half a step after u."""
        out = []
        for uid, iid in self._port_probe_ids:
            u, i = self.engine.get_probe(uid), self.engine.get_probe(iid)
            if allreduce is not None:
                u, i = allreduce(u), allreduce(i)
            out.append((u, i))
        return out

    def nf2ff_boxes(self, allreduce=None, freqs=None):
        """Frequency-domain surface data [nfreq][k][j][i] per recording request.  `freqs`: recorder mode only — any
        frequencies up to nf2ff_fmax (default: nf2ff_freqs); in dft mode the recorded set is returned."""
        if self.nf2ff_mode == "record":
            f = self.nf2ff_freqs if freqs is None else np.atleast_1d(np.asarray(freqs, float))
            if f.size and float(np.max(f)) > self.nf2ff_fmax * (1 + 1e-9):
                raise ValueError(f"NF2FF frequency {float(np.max(f)):g} Hz is above the recorder's band ({self.nf2ff_fmax:g} Hz)")
            tw_v = dft_twiddles(f, self.dt, self.dft_every, self.dft_nsamples, 0.0)
            tw_i = dft_twiddles(f, self.dt, self.dft_every, self.dft_nsamples, 0.5)
            boxes = self.nf2ff_box.collect(self.engine, self._nf_ids, tw_v, tw_i)
        else:
            if freqs is not None:
                raise ValueError("dft mode records nf2ff_freqs only")
            boxes = self.nf2ff_box.collect(self.engine, self._nf_ids)
        if allreduce is not None:
            boxes = [allreduce(b) for b in boxes]
        # single-sided spectra, as the port spectra of LumpedPort.CalcPort (2 dt sum u exp(-jwt): [EXT] openEMS's convention for both): with the
        # factor 2 on one side only, Prad / P_acc of a loss-free antenna reads 25 % (tests/test_tutorial_kat_cpu.py: power balance)
        scale = 2.0 * self.dt * self.dft_every
        return [b * scale for b in boxes]

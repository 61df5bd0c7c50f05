"""Solver plugin surface of the MI355X backend: probe_* / prepare_* / run_prepared_*.

Drop-in counterpart of the reference's ``antenna_sim/solver_fdtd_openems*.py`` family — same
three-call protocol, keyword arguments and result shape:

    probe_hip(dll_dir)                                   <- probe_openems_fixed          fixed.py:92
    prepare_hip_patch_fixed(params, dll_dir=...)         <- prepare_openems_patch_fixed  fixed.py:113
    prepare_hip_microstrip_patch(...)                    <- prepare_openems_microstrip_patch      microstrip.py:134
    prepare_hip_microstrip_patch_3d(...)                 <- prepare_openems_microstrip_patch_3d   microstrip_3d.py:19
    prepare_hip_microstrip_multi_3d(patches, ...)        <- prepare_openems_microstrip_multi_3d   multi_3d.py:98
    prepare_hip_patch(params, ...)                       <- prepare_openems_patch (legacy)        openems.py:140
    run_prepared_hip(prepared, frequency_hz=, verbose=)  <- run_prepared_openems_*  fixed.py:257, microstrip.py:369,
                                                            microstrip_3d.py:199, multi_3d.py:596, openems.py:271

The far field is evaluated at ``frequency_hz`` for every variant, as the reference effectively does; the S11
resonance rule its microstrip variant spells out but never reaches (microstrip.py:393,407-433) is opt-in:
``prepared.pattern_at_resonance = True``.

``dll_dir`` is reinterpreted as the directory holding libfdtd_hip.so (None/"" = the in-tree build).
Functions never raise: failures come back as ``ok=False`` + message, like upstream
(fixed.py:253-254,341-342).  The scenes are written against ``openems_api`` (the mirror of the
openEMS/CSXCAD calls the reference makes) and are pinned call-for-call against the reference by
tests/test_plugin_surface_cpu.py with fixtures captured from it.

Additions over the reference's result type (compatible: extra optional fields): port time series,
S11(f) and the resonance pick specified by the reference's dead S11 block (microstrip.py:407-426),
run statistics (Mcells/s).
"""
from __future__ import annotations

import math
import os
import random
import time
from dataclasses import dataclass, field
from enum import Enum
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np

from .constants import C0, EPS0
from .params import PatchAntennaParams
from .patch_design import design_patch_for_frequency, calculate_microstrip_width
from . import openems_api as oa
from . import _capi


# ---------------------------------------------------------------------------------------------------
# data types (fixed.py:15-42, microstrip.py:29-34, multi_patch_designer.py:18-28)
# ---------------------------------------------------------------------------------------------------
class FeedDirection(str, Enum):
    POS_X = "+X"
    NEG_X = "-X"
    POS_Y = "+Y"
    NEG_Y = "-Y"


@dataclass
class PatchInstance:
    name: str
    params: PatchAntennaParams
    center_x_m: float = 0.0          # (defaults as upstream: multi_patch_designer.py:18-28)
    center_y_m: float = 0.0
    center_z_m: float = 0.0
    feed_direction: FeedDirection = FeedDirection.NEG_X
    rot_x_deg: float = 0.0
    rot_y_deg: float = 0.0
    rot_z_deg: float = 0.0


@dataclass
class FDTDProbe:
    ok: bool
    message: str
    api: Dict[str, List[str]] = field(default_factory=dict)


@dataclass
class FDTDPrepared:
    ok: bool
    message: str
    FDTD: Optional[object] = None
    nf: Optional[object] = None
    sim_path: Optional[str] = None
    theta: Optional[np.ndarray] = None      # degrees (radians for the legacy variant, as upstream)
    phi: Optional[np.ndarray] = None
    nf_center: Optional[np.ndarray] = None
    port: Optional[object] = None           # first port (upstream forgot to keep it: microstrip.py:393)
    ports: List[object] = field(default_factory=list)
    variant: str = "fixed"
    # True: far field at the S11 resonance instead of frequency_hz — what the reference's microstrip variant WRITES
    # (microstrip.py:407-433) but never DOES: its OpenEMSPrepared has no `port` field, so getattr(prepared, 'port', None) at
    # :393 is always None and f_res stays frequency_hz.  None / False = the reference's effective behaviour (every variant
    # evaluates at frequency_hz); the resonance rule is opt-in.
    pattern_at_resonance: Optional[bool] = None


@dataclass
class FDTDResult:
    ok: bool
    message: str
    theta: Optional[np.ndarray] = None      # radians
    phi: Optional[np.ndarray] = None
    intensity: Optional[np.ndarray] = None  # (ntheta, nphi) dBi
    sim_path: Optional[str] = None
    is_dBi: bool = False
    # --- extensions ---
    freq: Optional[np.ndarray] = None
    s11: Optional[np.ndarray] = None        # complex, first port
    s11_dB: Optional[np.ndarray] = None
    f_res: Optional[float] = None           # resonance pick of microstrip.py:414-423 (frequency_hz if no dip below -10 dB)
    f_pattern: Optional[float] = None       # frequency the far field was evaluated at
    Dmax: Optional[float] = None
    port_u: Optional[np.ndarray] = None
    port_i: Optional[np.ndarray] = None
    dt: Optional[float] = None
    stats: Optional[dict] = None


# upstream names, so `from ... import OpenEMSResult` style callers keep working
OpenEMSProbe, OpenEMSPrepared, OpenEMSResult = FDTDProbe, FDTDPrepared, FDTDResult


# ---------------------------------------------------------------------------------------------------
# helpers
# ---------------------------------------------------------------------------------------------------
def _load(dll_dir: Optional[str], backend: Optional[dict] = None):
    """libfdtd_hip.so from dll_dir (None = in-tree build); raises if missing — no CPU fallback.
    An explicit `lib=` backend option is an already loaded library exporting include/fdtd_hip.h (a build kept
    elsewhere; tests/ pass the CPU oracle, the checker, this way) — the same keyword openems_api.openEMS takes."""
    if backend is not None and backend.get("lib") is not None:
        return backend.pop("lib")
    if backend is not None:
        backend.pop("lib", None)
    return _capi.load_hip_library(dll_dir or None)


def _patch_dims_mm(p) -> tuple:
    """(W on x, L on y) in mm; designed for resonance unless both are given (fixed.py:141-149)."""
    if p.patch_length_m and p.patch_width_m:
        return p.patch_width_m * 1e3, p.patch_length_m * 1e3
    L, W, _ = design_patch_for_frequency(p.frequency_hz, p.eps_r, p.h_m)
    return W * 1e3, L * 1e3


def _unique_sim_path(work_dir: str) -> str:
    tag = time.strftime("%Y%m%d-%H%M%S") + f"-{os.getpid()}-{random.randint(1000, 9999)}"
    return os.path.abspath(f"{work_dir}_{tag}")


def _mesh_res(f0: float, fc: float, ppw: float, unit: float = 1e-3) -> float:
    return C0 / (f0 + fc) / unit / ppw


def _bc(boundary: str):
    return ["MUR"] * 6 if str(boundary).upper().startswith("MUR") else ["PML_8"] * 6


def _new_fdtd(nr_ts, end_criteria, f0, fc, bc, lib, backend):
    fdtd = oa.openEMS(NrTS=nr_ts, EndCriteria=end_criteria, lib=lib, **backend)
    fdtd.SetGaussExcite(f0, fc)
    fdtd.SetBoundaryCond(bc)
    csx = oa.ContinuousStructure()
    fdtd.SetCSX(csx)
    mesh = csx.GetGrid()
    mesh.SetDeltaUnit(1e-3)
    return fdtd, csx, mesh


_PPW_5 = {1: 12.0, 2: 16.0, 3: 20.0, 4: 25.0, 5: 32.0}
_PPW_10 = {**_PPW_5, 6: 40.0, 7: 50.0, 8: 65.0, 9: 80.0, 10: 100.0}


def _quality(q, hi) -> int:
    try:
        q = int(q)
    except Exception:
        q = 3
    return max(1, min(hi, q))


# ---------------------------------------------------------------------------------------------------
# probe
# ---------------------------------------------------------------------------------------------------
def probe_hip(dll_dir: Optional[str] = None) -> FDTDProbe:
    """Is the HIP library loadable and a GPU visible?  (probe_openems_fixed, fixed.py:92-110)"""
    api: Dict[str, List[str]] = {}
    try:
        lib = _load(dll_dir)
        api["libfdtd_hip"] = list(_capi.ABI_SYMBOLS)
        api["openEMS.openEMS"] = [n for n in dir(oa.openEMS) if not n.startswith("_")]
        api["CSXCAD.ContinuousStructure"] = [n for n in dir(oa.ContinuousStructure) if not n.startswith("_")]
        ndev = lib.fdtd_device_count()
        if ndev < 1:
            return FDTDProbe(False, "libfdtd_hip.so loaded but no HIP device is visible", api)
        return FDTDProbe(True, f"fdtd-hip backend ready: {lib.fdtd_backend().decode()}, {ndev} device(s), "
                               f"ABI v{lib.fdtd_version()} ({_capi.hip_library_path(dll_dir or None)})", api)
    except Exception as e:
        return FDTDProbe(False, f"fdtd-hip backend not available: {e}", api)


probe_hip_fixed = probe_hip_microstrip = probe_hip


# ---------------------------------------------------------------------------------------------------
# prepare: single patch, lumped port at x = -6 mm  (fixed.py:113-254)
# ---------------------------------------------------------------------------------------------------
def prepare_hip_patch_fixed(params, *, dll_dir: Optional[str] = None, work_dir: str = "fdtd_hip_out_fixed",
                            cleanup: bool = True, verbose: int = 0, **backend) -> FDTDPrepared:
    try:
        lib = _load(dll_dir, backend)
        f0 = params.frequency_hz
        fc = 0.5 * f0
        pw, pl = _patch_dims_mm(params)
        h = params.h_m * 1e3
        # upstream quirk kept on purpose: this variant scales kappa by an extra 1e-3 (fixed.py:153)
        kappa = 1e-3 * 2 * np.pi * f0 * EPS0 * params.eps_r * params.loss_tangent
        box = np.array([200, 200, 150])
        fdtd, csx, mesh = _new_fdtd(30000, 1e-4, f0, fc, ["MUR"] * 6, lib, backend)
        res = _mesh_res(f0, fc, 20)
        mesh.AddLine("x", [-box[0] / 2, box[0] / 2])
        mesh.AddLine("y", [-box[1] / 2, box[1] / 2])
        mesh.AddLine("z", [-box[2] / 3, box[2] * 2 / 3])
        patch = csx.AddMetal("patch")
        patch.AddBox(priority=10, start=[-pw / 2, -pl / 2, h], stop=[pw / 2, pl / 2, h])
        fdtd.AddEdges2Grid(dirs="xy", properties=patch, metal_edge_res=res / 2)
        sub = csx.AddMaterial("substrate", epsilon=params.eps_r, kappa=kappa)
        sub.AddBox(priority=0, start=[-30.0, -30.0, 0], stop=[30.0, 30.0, h])
        mesh.AddLine("z", np.linspace(0, h, 5))
        gnd = csx.AddMetal("gnd")
        gnd.AddBox([-30.0, -30.0, 0], [30.0, 30.0, 0], priority=10)
        fdtd.AddEdges2Grid(dirs="xy", properties=gnd)
        port = fdtd.AddLumpedPort(1, 50, [-6, 0, 0], [-6, 0, h], "z", 1.0, priority=5, edges2grid="xy")
        mesh.SmoothMeshLines("all", res, 1.4)
        nf = fdtd.CreateNF2FFBox()
        if verbose:
            print(f"[fdtd-hip] fixed scene: patch {pw:.1f} x {pl:.1f} mm, eps_r={params.eps_r:.2f}, h={h:.3f} mm, feed x=-6 mm")
        return FDTDPrepared(True, "Fixed solver prepared (fdtd-hip backend)", FDTD=fdtd, nf=nf,
                            sim_path=_unique_sim_path(work_dir), theta=np.arange(0.0, 180.0, 2.0), phi=[0.0, 90.0],
                            nf_center=np.array([0, 0, 1e-3]), port=port, ports=[port], variant="fixed")
    except Exception as e:
        return FDTDPrepared(False, f"Fixed solver prepare failed: {e}")


# ---------------------------------------------------------------------------------------------------
# prepare: microstrip-fed patch (microstrip.py:134-366) and its 3-D sampling sibling (microstrip_3d.py:19-196)
# ---------------------------------------------------------------------------------------------------
def _microstrip_scene(params, feed_direction, feed_len, boundary, air_margin, ppw, extra_feed_lines, lib, backend):
    f0 = params.frequency_hz
    fc = 0.5 * f0
    pw, pl = _patch_dims_mm(params)
    h = params.h_m * 1e3
    fw = calculate_microstrip_width(f0, params.eps_r, params.h_m) * 1e3
    fd = FeedDirection(getattr(feed_direction, "value", feed_direction))
    along_x = fd in (FeedDirection.POS_X, FeedDirection.NEG_X)
    sw = pw + 60.0 + (feed_len if along_x else 0.0)
    sl = pl + 60.0 + (0.0 if along_x else feed_len)
    bx, by, bz = sw + 2 * air_margin, sl + 2 * air_margin, 160.0
    fdtd, csx, mesh = _new_fdtd(30000, 1e-4, f0, fc, _bc(boundary), lib, backend)
    res = _mesh_res(f0, fc, ppw)
    mesh.AddLine("x", [-bx / 2, bx / 2])
    mesh.AddLine("y", [-by / 2, by / 2])
    mesh.AddLine("z", [-bz / 3, bz * 2 / 3])
    kappa = 2 * np.pi * f0 * EPS0 * params.eps_r * params.loss_tangent
    sub = csx.AddMaterial("substrate", epsilon=params.eps_r, kappa=kappa)
    sub.AddBox(priority=0, start=[-sw / 2, -sl / 2, 0], stop=[sw / 2, sl / 2, h])
    mesh.AddLine("z", np.linspace(0, h, 5))
    gnd = csx.AddMetal("ground")
    gnd.AddBox(priority=10, start=[-sw / 2, -sl / 2, 0], stop=[sw / 2, sl / 2, 0])
    fdtd.AddEdges2Grid(dirs="xy", properties=gnd)
    patch = csx.AddMetal("patch")
    patch.AddBox(priority=10, start=[-pw / 2, -pl / 2, h], stop=[pw / 2, pl / 2, h])
    fdtd.AddEdges2Grid(dirs="xy", properties=patch, metal_edge_res=res / 2)
    # feed strip from the substrate edge to the patch edge, and the feed point at the patch edge centre
    strip = {FeedDirection.NEG_X: ([-sw / 2, -fw / 2, h], [-pw / 2, fw / 2, h], (-pw / 2, 0.0)),
             FeedDirection.POS_X: ([pw / 2, -fw / 2, h], [sw / 2, fw / 2, h], (pw / 2, 0.0)),
             FeedDirection.NEG_Y: ([-fw / 2, -sl / 2, h], [fw / 2, -pl / 2, h], (0.0, -pl / 2)),
             FeedDirection.POS_Y: ([-fw / 2, pl / 2, h], [fw / 2, sl / 2, h], (0.0, pl / 2))}[fd]
    feed = csx.AddMetal("feed_line")
    feed.AddBox(priority=10, start=strip[0], stop=strip[1])
    fdtd.AddEdges2Grid(dirs="xy", properties=feed, metal_edge_res=res / 2)
    px, py = strip[2]
    mesh.AddLine("x", [float(px)])
    mesh.AddLine("y", [float(py)])
    mesh.AddLine("z", [0.0, float(h)])
    port = fdtd.AddLumpedPort(1, 50.0, [float(px), float(py), 0.0], [float(px), float(py), float(h)], "z", 1.0,
                              priority=5, edges2grid="xy")
    if extra_feed_lines:
        mesh.AddLine("y" if along_x else "x", [-fw / 2, 0, fw / 2])
    mesh.SmoothMeshLines("all", res, 1.4)
    nf = fdtd.CreateNF2FFBox()
    return fdtd, nf, port, h, fd


def prepare_hip_microstrip_patch(params, *, dll_dir: Optional[str] = None,
                                 feed_direction: FeedDirection = FeedDirection.NEG_X,
                                 feed_line_length_mm: float = 20.0, boundary: str = "MUR",
                                 theta_step_deg: float = 2.0, work_dir: str = "fdtd_hip_out_microstrip",
                                 cleanup: bool = True, verbose: int = 0, **backend) -> FDTDPrepared:
    try:
        lib = _load(dll_dir, backend)
        fdtd, nf, port, h, fd = _microstrip_scene(params, feed_direction, feed_line_length_mm, boundary, 50.0, 20,
                                                  True, lib, backend)
        theta = np.arange(0.0, 181.0, max(0.5, float(theta_step_deg)))
        return FDTDPrepared(True, f"Microstrip patch prepared (feed: {fd}, fdtd-hip backend)", FDTD=fdtd, nf=nf,
                            sim_path=_unique_sim_path(work_dir), theta=theta, phi=np.array([0.0, 90.0]),
                            nf_center=np.array([0.0, 0.0, h / 2000.0]), port=port, ports=[port], variant="microstrip")
    except Exception as e:
        return FDTDPrepared(False, f"Microstrip solver prepare failed: {e}")


def prepare_hip_microstrip_patch_3d(params, *, dll_dir: Optional[str] = None,
                                    feed_direction: FeedDirection = FeedDirection.NEG_X,
                                    feed_line_length_mm: float = 20.0, boundary: str = "MUR",
                                    theta_step_deg: float = 2.0, phi_step_deg: float = 5.0, mesh_quality: int = 3,
                                    work_dir: str = "fdtd_hip_out_microstrip", cleanup: bool = True, verbose: int = 0,
                                    **backend) -> FDTDPrepared:
    try:
        lib = _load(dll_dir, backend)
        ppw = _PPW_5[_quality(mesh_quality, 5)]
        fdtd, nf, port, h, fd = _microstrip_scene(params, feed_direction, feed_line_length_mm, boundary, 80.0, ppw,
                                                  False, lib, backend)
        theta = np.arange(0.0, 181.0, max(0.5, float(theta_step_deg)))
        phi = np.arange(0.0, 361.0, max(1.0, float(phi_step_deg)))
        return FDTDPrepared(True, "Microstrip 3D prepared", FDTD=fdtd, nf=nf, sim_path=_unique_sim_path(work_dir),
                            theta=theta, phi=phi, nf_center=np.array([0.0, 0.0, h / 2000.0]), port=port, ports=[port],
                            variant="microstrip_3d")
    except Exception as e:
        return FDTDPrepared(False, f"Microstrip 3D prepare failed: {e}")


# ---------------------------------------------------------------------------------------------------
# prepare: N rotated/translated patches, one lumped port each (multi_3d.py:98-593)
# ---------------------------------------------------------------------------------------------------
def _rot_rows(rx, ry, rz) -> np.ndarray:
    """Row-vector rotation world = local @ R for extrinsic X, then Y, then Z rotations (multi_3d.py:41-57)."""
    def r(axis, deg):
        a = math.radians(deg)
        c, s = math.cos(a), math.sin(a)
        m = np.eye(3)
        i, j = (axis + 1) % 3, (axis + 2) % 3
        m[i, i] = c; m[i, j] = -s; m[j, i] = s; m[j, j] = c
        return m
    return (r(2, rz) @ r(1, ry) @ r(0, rx)).T


def _placed(box, rx, ry, rz, T):
    for ax, ang in (("x", rx), ("y", ry), ("z", rz)):
        if abs(ang) > 1e-9:
            box.AddTransform("RotateAxis", ax, ang)
    box.AddTransform("Translate", T.tolist())


def prepare_hip_microstrip_multi_3d(patches: Sequence, *, dll_dir: Optional[str] = None, boundary: str = "MUR",
                                    theta_step_deg: float = 2.0, phi_step_deg: float = 5.0, mesh_quality: int = 3,
                                    nf_center_mode: str = "origin", simbox_mode: str = "auto",
                                    auto_margin_mm=(80.0, 80.0, 160.0), manual_size_mm=None,
                                    feed_line_length_mm: float = 20.0, port_mode: str = "lumped",
                                    end_criteria_db: float = -25.0, work_dir: str = "fdtd_hip_out_multi",
                                    cleanup: bool = True, verbose: int = 0, log_cb: Optional[Callable] = None,
                                    **backend) -> FDTDPrepared:
    try:
        if not patches:
            return FDTDPrepared(False, "No patch instances provided.")
        lib = _load(dll_dir, backend)

        def say(msg):
            try:
                (log_cb or print)(msg)
            except Exception:
                pass

        f0 = float(patches[0].params.frequency_hz)
        fc = 0.5 * f0
        # world bounds of all (rotated) substrates
        placed, pts, max_h = [], [], 0.0
        for inst in patches:
            pw, pl = _patch_dims_mm(inst.params)
            h = float(inst.params.h_m) * 1e3
            max_h = max(max_h, h)
            fd = FeedDirection(getattr(inst.feed_direction, "value", inst.feed_direction))
            along_x = fd in (FeedDirection.POS_X, FeedDirection.NEG_X)
            sw = pw + 60.0 + (feed_line_length_mm if along_x else 0.0)
            sl = pl + 60.0 + (0.0 if along_x else feed_line_length_mm)
            rot = tuple(float(getattr(inst, k, 0.0)) for k in ("rot_x_deg", "rot_y_deg", "rot_z_deg"))
            R = _rot_rows(*rot)
            T = np.array([inst.center_x_m, inst.center_y_m, inst.center_z_m], dtype=float) * 1e3
            corners = np.array([[sx * sw / 2, sy * sl / 2, sz * h / 2] for sz in (-1, 1) for sy in (-1, 1) for sx in (-1, 1)])
            pts.append(corners @ R + T)
            placed.append((inst, pw, pl, h, fd, sw, sl, rot, R, T))
        pts = np.concatenate(pts)
        lo, hi = pts.min(axis=0), pts.max(axis=0)
        ctr = 0.5 * (lo + hi)
        if str(simbox_mode or "auto").lower().startswith("man") and manual_size_mm is not None:
            size = np.array(manual_size_mm, dtype=float)
        else:
            size = (hi - lo) + 2.0 * np.array(auto_margin_mm, dtype=float)
        if verbose:
            say(f"SimBox (mm): X={size[0]:.1f} Y={size[1]:.1f} Z={size[2]:.1f}")
        q = _quality(mesh_quality, 10)
        ppw = _PPW_10[q]
        res = _mesh_res(f0, fc, ppw)
        nr_ts = {6: 50000, 7: 70000, 8: 100000, 9: 130000, 10: 160000}.get(q, 30000)
        # thin copper forces a tiny time step: keep the whole excitation pulse inside NrTS (multi_3d.py:244-270)
        t_cu = [max(0.02, float(inst.params.metal.thickness_m) * 1e3) for inst in patches]
        min_dim_m = min(min(t_cu), float(res)) * 1e-3
        exc_ts = max(10000, int((3.35 / fc) / max(1e-15, min_dim_m / (C0 * 1.8))))
        nr_ts = max(nr_ts, min(220000, int(2.2 * exc_ts)))
        try:
            ec_db = float(end_criteria_db)
        except Exception:
            ec_db = -25.0
        ec_db = max(-80.0, min(-10.0, ec_db))
        if verbose:
            say(f"Mesh: q={q} -> ppw={ppw:g}, mesh_res={res:.3f} mm, NrTS={nr_ts}; EndCriteria={ec_db:g} dB")
        fdtd, csx, mesh = _new_fdtd(nr_ts, 10.0 ** (ec_db / 20.0), f0, fc, _bc(boundary), lib, backend)
        for a, ax in enumerate("xyz"):
            mesh.AddLine(ax, [ctr[a] - size[a] / 2, ctr[a] + size[a] / 2])

        def plane_lines(corners_local, R, T, density):
            w = np.asarray(corners_local, dtype=float) @ R + T
            n = max(3, int(density))
            for a, ax in enumerate("xyz"):
                mesh.AddLine(ax, np.linspace(float(w[:, a].min()), float(w[:, a].max()), n).tolist())

        ports = []
        for idx, (inst, pw, pl, h, fd, sw, sl, rot, R, T) in enumerate(placed, start=1):
            p = inst.params
            fw = calculate_microstrip_width(p.frequency_hz, p.eps_r, p.h_m) * 1e3
            kappa = 2 * np.pi * p.frequency_hz * EPS0 * p.eps_r * p.loss_tangent
            sub = csx.AddMaterial(f"substrate_{idx}", epsilon=p.eps_r, kappa=kappa)
            m_gnd, m_patch, m_feed = (csx.AddMetal(f"{n}_{idx}") for n in ("ground", "patch", "feed"))
            t = max(0.02, float(p.metal.thickness_m) * 1e3)
            _placed(sub.AddBox(priority=0, start=[-sw / 2, -sl / 2, -h / 2], stop=[sw / 2, sl / 2, h / 2]), *rot, T)
            normal = np.array([0.0, 0.0, 1.0]) @ R
            th_axis = int(np.argmax(np.abs(normal)))
            mesh.AddLine("xyz"[th_axis], np.linspace(float(T[th_axis]) - h / 2, float(T[th_axis]) + h / 2, 5).tolist())
            _placed(m_gnd.AddBox(priority=10, start=[-sw / 2, -sl / 2, -h / 2 - t / 2], stop=[sw / 2, sl / 2, -h / 2 + t / 2]), *rot, T)
            plane_lines([[sx * sw / 2, sy * sl / 2, -h / 2] for sx, sy in ((-1, -1), (1, -1), (1, 1), (-1, 1))], R, T, 6 + 2 * q)
            _placed(m_patch.AddBox(priority=10, start=[-pw / 2, -pl / 2, h / 2 - t / 2], stop=[pw / 2, pl / 2, h / 2 + t / 2]), *rot, T)
            plane_lines([[sx * pw / 2, sy * pl / 2, h / 2] for sx, sy in ((-1, -1), (1, -1), (1, 1), (-1, 1))], R, T, 6 + 2 * q)
            fx, fy = {FeedDirection.NEG_X: (-pw / 2, 0.0), FeedDirection.POS_X: (pw / 2, 0.0),
                      FeedDirection.NEG_Y: (0.0, -pl / 2), FeedDirection.POS_Y: (0.0, pl / 2)}[fd]
            pad = max(1.0, float(fw))
            p0 = [fx - pad / 2, fy - pad / 2, h / 2 - t / 2]
            p1 = [fx + pad / 2, fy + pad / 2, h / 2 + t / 2]
            _placed(m_feed.AddBox(priority=11, start=p0, stop=p1), *rot, T)
            plane_lines([[p0[0], p0[1], h / 2], [p1[0], p0[1], h / 2], [p1[0], p1[1], h / 2], [p0[0], p1[1], h / 2]], R, T, 8 + 2 * q)
            # lumped port box from ground to patch along the world axis closest to the substrate normal
            normal = normal / max(1e-12, np.linalg.norm(normal))
            absn = np.abs(normal)
            axis = int(np.argmax(absn))
            if abs(absn[2] - absn[axis]) < 1e-6:
                axis = 2
            c_w = (np.array([fx, fy, h / 2]) @ R + T)
            g_w = (np.array([fx, fy, -h / 2]) @ R + T)
            ext = max(0.1, 0.25 * res)
            a_lo, a_hi = sorted([g_w[axis], c_w[axis]])
            a_lo, a_hi = float(a_lo - ext), float(a_hi + ext)
            half = max(0.4, min(0.6 * float(fw), 0.35 * float(res)))
            s0, s1 = (a_lo, a_hi) if normal[axis] >= 0.0 else (a_hi, a_lo)
            start = [float(c_w[0] - half), float(c_w[1] - half), float(c_w[2] - half)]
            stop = [float(c_w[0] + half), float(c_w[1] + half), float(c_w[2] + half)]
            start[axis], stop[axis] = s0, s1
            for a in [axis] + [b for b in (0, 1, 2) if b != axis]:
                mesh.AddLine("xyz"[a], [start[a], float(c_w[a]), stop[a]])
            if verbose:
                say(f"Patch {idx}: center(mm)={np.round(T, 3).tolist()} rot={rot} port axis={axis} "
                    f"start={np.round(start, 2).tolist()} stop={np.round(stop, 2).tolist()}")
            ports.append(fdtd.AddLumpedPort(idx, 50.0, start, stop, axis, excite=+1, priority=5,
                                            edges2grid="".join(c for i, c in enumerate("xyz") if i != axis)))
        mesh.SmoothMeshLines("all", res, 1.4)
        nf = fdtd.CreateNF2FFBox()
        if str(nf_center_mode or "origin").lower().startswith("cent"):
            cen = np.array([float(np.mean([p.center_x_m for p in patches])) * 1e3,
                            float(np.mean([p.center_y_m for p in patches])) * 1e3,
                            float(np.mean([p.center_z_m for p in patches])) * 1e3 + max_h / 2000.0])
        else:
            cen = np.array([0.0, 0.0, max_h / 2000.0])
        theta = np.arange(0.0, 181.0, max(0.5, float(theta_step_deg)))
        phi = np.arange(0.0, 361.0, max(1.0, float(phi_step_deg)))
        return FDTDPrepared(True, "Microstrip multi-antenna 3D prepared", FDTD=fdtd, nf=nf,
                            sim_path=_unique_sim_path(work_dir), theta=theta, phi=phi, nf_center=cen,
                            port=ports[0], ports=ports, variant="multi_3d")
    except Exception as e:
        return FDTDPrepared(False, f"Microstrip multi-3D prepare failed: {e}")


# ---------------------------------------------------------------------------------------------------
# prepare: legacy full-3D variant (openems.py:140-268)
# ---------------------------------------------------------------------------------------------------
def prepare_hip_patch(params, *, dll_dir: Optional[str] = None, work_dir: str = "fdtd_hip_out", cleanup: bool = True,
                      verbose: int = 0, **backend) -> FDTDPrepared:
    try:
        lib = _load(dll_dir, backend)
        f0 = params.frequency_hz
        fc = 0.5 * f0
        W, L = _patch_dims_mm(params)
        h = params.h_m * 1e3
        res = 299792458.0 / (f0 + fc) / 1e-3 / 20.0
        fdtd, csx, mesh = _new_fdtd(60000, 1e-5, f0, fc, [3] * 6, lib, backend)
        mesh.AddLine("x", [-100.0, 100.0])
        mesh.AddLine("y", [-100.0, 100.0])
        mesh.AddLine("z", [-50.0, 100.0])
        kappa = 2.0 * np.pi * f0 * 8.854187817e-12 * params.eps_r * max(0.0, params.loss_tangent)   # openems.py:199-200
        sub = csx.AddMaterial("substrate", epsilon=params.eps_r, kappa=kappa)
        sub.AddBox([-100.0, -100.0, 0.0], [100.0, 100.0, h])
        mesh.AddLine("z", np.linspace(0.0, h, 5).tolist())
        gnd = csx.AddMetal("gnd")
        gnd.AddBox([-100.0, -100.0, 0.0], [100.0, 100.0, 0.0], priority=10)
        patch = csx.AddMetal("patch")
        patch.AddBox([-W / 2.0, -L / 2.0, h], [W / 2.0, L / 2.0, h], priority=10)
        fdtd.AddEdges2Grid(dirs="xy", properties=patch, metal_edge_res=res / 2.0)
        fdtd.AddEdges2Grid(dirs="xy", properties=gnd)
        fx = -0.2 * W
        mesh.AddLine("x", [float(fx)])
        mesh.AddLine("z", [0.0, float(h)])
        port = fdtd.AddLumpedPort(1, 50, [float(fx), 0.0, 0.0], [float(fx), 0.0, float(h)], "z", 1.0, priority=5,
                                  edges2grid="xy")
        mesh.SmoothMeshLines("all", res, 1.4)
        nf = fdtd.CreateNF2FFBox()
        return FDTDPrepared(True, "Prepared (fdtd-hip backend)", FDTD=fdtd, nf=nf, sim_path=os.path.abspath(work_dir),
                            theta=np.linspace(0, np.pi, 91), phi=np.linspace(0, 2 * np.pi, 181),
                            nf_center=np.array([0.0, 0.0, 1e-3]), port=port, ports=[port], variant="legacy")
    except Exception as e:
        return FDTDPrepared(False, f"prepare failed: {e}")


# ---------------------------------------------------------------------------------------------------
# run + result conversion
# ---------------------------------------------------------------------------------------------------
def pattern_to_dBi(E_norm: np.ndarray, Dmax: Optional[float], variant: str = "fixed") -> np.ndarray:
    """The only arithmetic the reference owns (fixed.py:309-315; microstrip.py:454-455;
    microstrip_3d.py:240-248; multi_3d.py:635-642): 20 log10(E/Emax [+1e-16]) + 10 log10(Dmax).
    (The legacy variant converts differently: legacy_pattern.)"""
    E = np.asarray(E_norm, dtype=float)
    e_max = float(np.max(E)) if E.size else 1.0
    if variant in ("microstrip_3d", "multi_3d"):
        if e_max <= 0:
            e_max = 1.0
        out = 20.0 * np.log10(E / e_max + 1e-16)
        return out + 10.0 * np.log10(Dmax) if (Dmax is not None and Dmax > 0) else out
    if variant == "fixed" and e_max <= 0:
        return np.full_like(E, -50.0)
    return 20.0 * np.log10(E / e_max) + 10 * np.log10(Dmax)


def legacy_pattern(res, n_theta: int, n_phi: int):
    """(intensity (n_theta, n_phi), is_dBi) of the LEGACY variant, which sniffs the attributes of the NF2FF result
    (solver_fdtd_openems.py:307-408): directivity 4 pi P_rad / Prad in dBi when both are there; if that comes out below
    -10 dBi everywhere ("obviously wrong") E_norm + Dmax as the tutorials do; without P_rad / Prad the normalised
    |E_theta|^2 + |E_phi|^2 (or any magnitude array), linear, is_dBi False.  Then the grid is forced to (n_theta, n_phi)."""
    def first(*names):
        for n in names:
            if hasattr(res, n):
                return getattr(res, n)
        return None

    e_th, e_ph = first("E_theta", "Eth", "E_th", "Etheta"), first("E_phi", "Eph", "E_ph", "Ephi")
    p_rad, prad = first("P_rad", "P", "U"), first("Prad", "Ptot", "P_rad_tot")
    arr, dbi = None, False
    if p_rad is not None and prad is not None:
        try:
            g = np.asarray(p_rad)
            g = np.asarray(g[0] if g.ndim == 3 else g, dtype=float)
            arr = 10.0 * np.log10(np.maximum(1e-16, g / max(1e-16, float(np.asarray(prad).flat[0])) * (4.0 * np.pi)))
            dbi = True
        except Exception:
            arr, dbi = None, False
    if dbi and (np.nanmax(arr) < -10.0 or np.isnan(np.nanmax(arr))):
        e_norm, dmax = first("E_norm"), first("Dmax")
        if e_norm is not None and dmax is not None:
            try:
                g = np.asarray(e_norm)
                g = np.array(g[0] if g.ndim == 3 else g, dtype=float)
                g /= max(1e-16, float(g.max()))
                arr = 20.0 * np.log10(np.maximum(1e-16, g)) + 10.0 * np.log10(max(1e-16, float(np.asarray(dmax).flat[0])))
            except Exception:
                pass
    if arr is None:
        if e_th is None or e_ph is None:
            u = first("U", "Gain", "E_norm")
            if u is None:
                raise ValueError("NF2FF result has no usable field data")
            lin = np.abs(u).astype(float)
        else:
            lin = (np.abs(e_th) ** 2 + np.abs(e_ph) ** 2).astype(float)
        arr, dbi = lin / max(1e-16, float(np.max(lin))), False
    n = n_theta * n_phi
    if arr.size == n:
        arr = arr.reshape(n_theta, n_phi)
    elif arr.size > n:
        arr = arr.flat[:n].reshape(n_theta, n_phi)
    else:
        pad = np.zeros((n_theta, n_phi))
        pad.flat[:arr.size] = arr.flat
        arr = pad
    arr = np.asarray(arr, dtype=float)
    if not dbi:
        arr = arr / max(1e-16, float(arr.max()))
    return arr, dbi


def s11_from_port(port, sim_path, f_center: float, npts: int = 201):
    """S11 over 0.7..1.3 f (>= 1 GHz) and the resonance pick of microstrip.py:407-426:
    the minimum if it is below -10 dB, else the requested frequency."""
    f = np.linspace(max(1e9, f_center * 0.7), f_center * 1.3, npts)
    port.CalcPort(sim_path, f)
    s11 = port.uf_ref / port.uf_inc
    s11_dB = 20.0 * np.log10(np.abs(s11))
    idx = np.where((s11_dB < -10) & (s11_dB == np.min(s11_dB)))[0]
    f_res = float(f[idx[0]]) if len(idx) == 1 else float(f_center)
    return f, s11, s11_dB, f_res


def _merged_lines(fdtd) -> dict:
    try:
        return {"xyz"[a]: list(m) for a, m in enumerate(fdtd.GetCSX().GetGrid().merged_lines) if m}
    except AttributeError:      # an engine object without this package's mesh (duck-typed FDTD objects)
        return {}


def run_prepared_hip(prepared: FDTDPrepared, *, frequency_hz: float, verbose: int = 1) -> FDTDResult:
    """Time-step on the GPU, then far field at `frequency_hz` on the prepared theta x phi grid in ONE
    transform (the reference loops CalcNF2FF over phi: microstrip_3d.py:224-238)."""
    try:
        if not prepared.ok or prepared.FDTD is None or prepared.nf is None:
            return FDTDResult(False, prepared.message)
        fdtd, nf = prepared.FDTD, prepared.nf
        sim_path = prepared.sim_path or "fdtd_hip_out"
        if verbose:
            print(f"[fdtd-hip] starting FDTD ({prepared.variant}) in: {sim_path}", flush=True)
        legacy = prepared.variant == "legacy"
        fdtd.Run(sim_path, verbose=verbose, cleanup=not legacy)      # (the legacy variant keeps a pre-existing sim_path: openems.py:289)
        th = np.asarray(prepared.theta, dtype=float)
        ph = np.asarray(prepared.phi, dtype=float)
        th_deg, ph_deg = (np.rad2deg(th), np.rad2deg(ph)) if legacy else (th, ph)
        # port parameters first (microstrip.py:407-426): the resonance they give is where the microstrip variant
        # takes its far field (:433)
        s11_out = s11_from_port(prepared.port, sim_path, frequency_hz) if prepared.port is not None else None
        at_res = bool(prepared.pattern_at_resonance)     # None: as the reference effectively does — at frequency_hz
        f_eval = float(s11_out[3]) if (at_res and s11_out is not None) else float(frequency_hz)
        if f_eval != float(frequency_hz) and not nf.can_evaluate(f_eval):
            f_eval = float(frequency_hz)                 # running-DFT faces at caller-named frequencies: the resonance was not recorded
        res = nf.CalcNF2FF(sim_path, f_eval, th_deg, ph_deg, center=prepared.nf_center)
        Dmax = float(np.asarray(res.Dmax)[0])
        if legacy:      # attribute-sniffing conversion; theta / phi go out as they came in (radians), openems.py:408
            intensity, is_dbi = legacy_pattern(res, th.size, ph.size)
            th_out, ph_out = th, ph
        else:
            e_norm = np.asarray(res.E_norm[0])
            d_ref = Dmax
            if prepared.variant in ("microstrip_3d", "multi_3d") and e_norm.ndim == 2 and e_norm.size and float(e_norm.max()) > 0:
                # The reference calls CalcNF2FF once per phi and keeps the Dmax of the FIRST call (microstrip_3d.py:224-237, multi_3d.py:620-633:
                # "expected to be constant") — but a call's Dmax is 4 pi max(P_rad) / Prad over the angles of THAT call, the phi[0] cut.  One
                # transform of the whole grid here; the same number: Dmax scaled by the cut's share of the peak.  (Equal whenever the
                # peak is at theta = 0, which lies in every cut; a two-element array peaked off broadside by 0.14 dB.)
                d_ref = Dmax * (float(e_norm[:, 0].max()) / float(e_norm.max())) ** 2
            intensity, is_dbi = pattern_to_dBi(e_norm, d_ref, prepared.variant), True
            th_out, ph_out = np.deg2rad(th_deg), np.deg2rad(ph_deg)
        out = FDTDResult(True, f"fdtd-hip FDTD completed ({prepared.variant})", theta=th_out,
                         phi=ph_out, intensity=intensity, sim_path=sim_path, is_dBi=is_dbi, Dmax=Dmax)
        out.f_pattern = float(np.atleast_1d(res.freq)[0])     # == f_eval unless a dft-mode comb snapped it
        if s11_out is not None:
            out.freq, out.s11, out.s11_dB, out.f_res = s11_out
            out.port_u = prepared.port.u_data.ui_val[0]
            out.port_i = prepared.port.i_data.ui_val[0]
            out.dt = fdtd.sim.dt
        st = fdtd.stats
        out.stats = {"steps": st.steps, "seconds": st.seconds, "mcells_per_s": st.mcells_per_s,
                     "energy_db": float(st.energy_db), "cells": fdtd.sim.grid.ncells,
                     "grid": list(fdtd.sim.grid.shape), "schedule_fallback": getattr(st, "schedule_fallback", None),
                     "halo_transports_failed": list(getattr(st, "transports_failed", ())),
                     "nf2ff_warning": getattr(fdtd.sim, "nf2ff_warning", None),
                     "excitation_warning": getattr(fdtd.sim, "excitation_warning", None),
                     # hint-line pairs the mesher merged (mesher.merge_close_lines): where this mesh differs from the one openEMS would build
                     "mesh_lines_merged": _merged_lines(fdtd)}
        if verbose:
            print(f"[fdtd-hip] done: {st.steps} steps, {st.mcells_per_s:.0f} MC/s, Dmax {10 * np.log10(Dmax):.2f} dBi", flush=True)
        return out
    except Exception as e:
        return FDTDResult(False, f"fdtd-hip run failed: {e}")


run_prepared_hip_fixed = run_prepared_hip_microstrip = run_prepared_hip_microstrip_3d = run_prepared_hip
run_prepared_hip_microstrip_multi_3d = run_prepared_hip_legacy = run_prepared_hip

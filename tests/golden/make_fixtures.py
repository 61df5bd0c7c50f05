#!/usr/bin/env python3
"""Generates the boundary golden fixtures in this directory by IMPORTING the reference
(/root/reference, read-only, this container only) against a recording fake of the external
openEMS/CSXCAD modules it drives.  Nothing of the reference's source is copied: the outputs are
data — the exact call sequence each ``prepare_*`` emits for given inputs (boxes, materials, mesh
hint lines, ports, NrTS, EndCriteria, f0/fc, boundary, theta/phi, nf_center), the values of its
closed-form design helpers, and ``run_prepared_*`` outputs for synthetic NF2FF results.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_fixtures.py

The FDTD engine itself (openEMS) is absent, so these fixtures pin the drop-in BOUNDARY
(SURVEY §8b/§8f), not the field solve (parity unpinned, see oracle/fdtd_oracle.c).
"""
import json
import os
import sys
import tempfile
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True


def _f(v):
    if isinstance(v, (list, tuple, np.ndarray)):
        return [_f(x) for x in np.asarray(v).tolist()] if not isinstance(v, (list, tuple)) else [_f(x) for x in v]
    if isinstance(v, (np.floating, float)):
        return float(v)
    if isinstance(v, (np.integer, int)):
        return int(v)
    return v


class Recorder:
    def __init__(self):
        self.calls = []

    def add(self, op, **kw):
        self.calls.append({"op": op, **{k: _f(v) for k, v in kw.items()}})


def legacy_synthetic(th, ph):
    """Synthetic far field for the legacy variant's conversion: complex E_theta / E_phi on the theta x phi grid (radians,
    broadcast shapes), P_rad = |E|^2 / (2 eta0), Prad = its integral over the sphere (rectangle rule), Dmax = 4 pi max(P_rad) / Prad.
    tests/test_plugin_surface_cpu.py builds the same arrays to check this package's conversion against the fixture."""
    e_th = np.cos(th / 2.0) ** 2 * (1.0 + 0.25 * np.cos(ph)) * np.exp(0.3j) + 0j
    e_ph = 0.3 * np.sin(th) * np.sin(ph) * np.exp(-0.7j) + 0j
    p_rad = (np.abs(e_th) ** 2 + np.abs(e_ph) ** 2) / (2.0 * 376.730313668)
    dth = float(th[1, 0] - th[0, 0]) if th.shape[0] > 1 else 1.0
    dph = float(ph[0, 1] - ph[0, 0]) if ph.shape[1] > 1 else 1.0
    prad = float(np.sum(p_rad * np.sin(th)) * dth * dph)
    return e_th, e_ph, p_rad, prad, float(4.0 * np.pi * p_rad.max() / prad)


def install_fake(rec: Recorder):
    class Box:
        def __init__(self, entry):
            self.entry = entry

        def AddTransform(self, kind, *args):
            self.entry.setdefault("transforms", []).append([kind] + [_f(a) for a in args])

    class Prop:
        def __init__(self, kind, name, **kw):
            self.kind, self.name = kind, name
            rec.add("Add" + kind, name=name, **kw)

        def AddBox(self, start=None, stop=None, priority=0, **kw):
            rec.add("AddBox", prop=self.name, priority=priority, start=list(start), stop=list(stop))
            return Box(rec.calls[-1])

    class Grid:
        def SetDeltaUnit(self, u):
            rec.add("SetDeltaUnit", unit=u)

        def AddLine(self, axis, lines):
            rec.add("AddLine", axis=axis, lines=np.atleast_1d(np.asarray(lines, float)))

        def SmoothMeshLines(self, axis, max_res, ratio=1.5):
            rec.add("SmoothMeshLines", axis=axis, max_res=max_res, ratio=ratio)

    class CSX:
        def __init__(self):
            self.grid = Grid()

        def GetGrid(self):
            return self.grid

        def AddMaterial(self, name, **kw):
            return Prop("Material", name, **kw)

        def AddMetal(self, name):
            return Prop("Metal", name)

    class NF2FF:
        # what the synthetic result object carries: "enorm" (E_norm + Dmax: all the newer variants read), or, for the
        # legacy variant's attribute-sniffing conversion (solver_fdtd_openems.py:307-408), "full" (every attribute the
        # nf2ff tool gives), "weak" (the same with P_rad a factor 1e-4 too small, so that the directivity from
        # P_rad / Prad looks "obviously wrong" and the E_norm + Dmax branch takes over), "fields" (E_theta, E_phi only)
        mode = "enorm"

        def CalcNF2FF(self, sim_path, freq, theta, phi, center=None, **kw):
            theta = np.atleast_1d(np.asarray(theta, float)); phi = np.atleast_1d(np.asarray(phi, float))
            rec.add("CalcNF2FF", freq=freq, ntheta=int(theta.size), nphi=int(phi.size), center=center,
                    theta_first_last=[float(theta[0]), float(theta[-1])], phi_first_last=[float(phi[0]), float(phi[-1])])
            th = np.deg2rad(theta)[:, None]; ph = np.deg2rad(phi)[None, :]
            res = types.SimpleNamespace()
            if NF2FF.mode == "enorm":
                res.E_norm = [np.cos(th / 2.0) ** 2 * (1.0 + 0.25 * np.cos(ph)) + 1e-3]
                res.Dmax = [4.0]
                return res
            e_th, e_ph, p_rad, prad, dmax = legacy_synthetic(th, ph)
            if NF2FF.mode in ("full", "weak"):
                res.E_theta, res.E_phi = [e_th], [e_ph]
                res.E_norm = [np.sqrt(np.abs(e_th) ** 2 + np.abs(e_ph) ** 2)]
                res.P_rad = [p_rad * (1e-4 if NF2FF.mode == "weak" else 1.0)]
                res.Prad, res.Dmax = [prad], [dmax]
            else:
                res.E_theta, res.E_phi = e_th, e_ph
            return res

    install_fake.NF2FF = NF2FF

    class FDTD:
        def __init__(self, NrTS=None, EndCriteria=None, **kw):
            rec.add("openEMS", NrTS=NrTS, EndCriteria=EndCriteria)

        def SetGaussExcite(self, f0, fc):
            rec.add("SetGaussExcite", f0=f0, fc=fc)

        def SetBoundaryCond(self, bc):
            rec.add("SetBoundaryCond", bc=list(bc))

        def SetCSX(self, csx):
            pass

        def AddEdges2Grid(self, dirs, properties=None, metal_edge_res=None, **kw):
            rec.add("AddEdges2Grid", dirs=dirs, prop=getattr(properties, "name", None), metal_edge_res=metal_edge_res)

        def AddLumpedPort(self, port_nr, R, start, stop, p_dir, excite=0, priority=0, edges2grid=None, **kw):
            rec.add("AddLumpedPort", port_nr=port_nr, R=R, start=list(start), stop=list(stop), p_dir=p_dir,
                    excite=excite, priority=priority, edges2grid=edges2grid)
            return object()

        def CreateNF2FFBox(self, **kw):
            rec.add("CreateNF2FFBox")
            return NF2FF()

        def Run(self, sim_path, verbose=0, cleanup=False, **kw):
            rec.add("Run", verbose=verbose, cleanup=cleanup)

    oem = types.ModuleType("openEMS")
    oem.openEMS = FDTD
    pc = types.ModuleType("openEMS.physical_constants")
    pc.C0 = 299792458.0
    pc.MUE0 = 4e-7 * np.pi
    pc.EPS0 = 1.0 / (pc.MUE0 * pc.C0 ** 2)
    oem.physical_constants = pc
    csx = types.ModuleType("CSXCAD")
    csx.ContinuousStructure = CSX
    oem.CSXCAD = csx
    sys.modules["openEMS"] = oem
    sys.modules["openEMS.physical_constants"] = pc
    sys.modules["CSXCAD"] = csx
    if not hasattr(os, "add_dll_directory"):
        os.add_dll_directory = lambda p: None


def main():
    sys.path.insert(0, REF)
    rec = Recorder()
    install_fake(rec)
    dll = tempfile.mkdtemp(prefix="fake_openems_")
    open(os.path.join(dll, "openEMS.dll"), "w").close()
    work = tempfile.mkdtemp(prefix="fake_work_")

    from antenna_sim.models import PatchAntennaParams
    from antenna_sim.physics import design_patch_for_frequency, effective_eps, delta_L
    from antenna_sim import solver_fdtd_openems_fixed as fx
    from antenna_sim import solver_fdtd_openems_microstrip as ms
    from antenna_sim import solver_fdtd_openems_microstrip_3d as m3
    from antenna_sim import solver_fdtd_openems_microstrip_multi_3d as mm
    from antenna_sim import solver_fdtd_openems as lg

    def PatchInstance(**kw):
        # multi_patch_designer.PatchInstance (multi_patch_designer.py:18-28) is a plain dataclass inside a
        # Tk module (tkinter is not installed here); the solver only duck-types it (PatchLike,
        # solver_fdtd_openems_microstrip_multi_3d.py:20-32), so a namespace with the same fields serves.
        base = dict(rot_x_deg=0.0, rot_y_deg=0.0, rot_z_deg=0.0)
        base.update(kw)
        return types.SimpleNamespace(**base)

    # ---- closed-form helpers --------------------------------------------------------------------
    design = []
    for f, er, h in [(2.45e9, 4.3, 1.6e-3), (5.8e9, 4.3, 1.6e-3), (2.0e9, 3.38, 1.524e-3), (1.0e9, 2.2, 0.787e-3),
                     (10e9, 9.8, 0.635e-3)]:
        L, W, ee = design_patch_for_frequency(f, er, h)
        design.append({"f": f, "eps_r": er, "h": h, "L": L, "W": W, "eps_eff": ee,
                       "dL": delta_L(ee, h, W), "eps_eff_direct": effective_eps(er, h, W),
                       "w50": float(ms.calculate_microstrip_width(f, er, h)),
                       "w30": float(ms.calculate_microstrip_width(f, er, h, 30.0)),
                       "w75": float(ms.calculate_microstrip_width(f, er, h, 75.0))})
    json.dump(design, open(os.path.join(OUT, "design_values.json"), "w"), indent=1)

    def params(f=2.45e9, **kw):
        return PatchAntennaParams.from_user_units(frequency_ghz=f / 1e9, er=4.3, h_mm=1.6, loss_tangent=0.02, **kw)

    def capture(fn, *a, **kw):
        rec.calls = []
        prep = fn(*a, **kw)
        assert prep.ok, prep.message
        return {"calls": rec.calls, "theta": _f(np.asarray(prep.theta)), "phi": _f(np.asarray(prep.phi)),
                "nf_center": _f(np.asarray(prep.nf_center)), "message_ok": True}, prep

    scenes = {}
    scenes["fixed_2g45"], prep_fixed = capture(fx.prepare_openems_patch_fixed, params(), dll_dir=dll, work_dir=os.path.join(work, "a"))
    scenes["fixed_explicit_LW"], _ = capture(fx.prepare_openems_patch_fixed, params(L_mm=28.0, W_mm=36.0), dll_dir=dll, work_dir=os.path.join(work, "a"))
    for name, fd in (("negx", ms.FeedDirection.NEG_X), ("posy", ms.FeedDirection.POS_Y)):
        scenes[f"microstrip_{name}"], prep_ms = capture(ms.prepare_openems_microstrip_patch, params(), dll_dir=dll, feed_direction=fd,
                                                        boundary="MUR", theta_step_deg=2.0, work_dir=os.path.join(work, "b"))
    scenes["microstrip3d_5g8_pml_q3"], prep_3d = capture(m3.prepare_openems_microstrip_patch_3d, params(5.8e9), dll_dir=dll,
                                                         feed_direction=ms.FeedDirection.NEG_X, boundary="PML_8", theta_step_deg=2.0,
                                                         phi_step_deg=5.0, mesh_quality=3, work_dir=os.path.join(work, "c"))
    scenes["microstrip3d_2g45_mur_q5_posx"], _ = capture(m3.prepare_openems_microstrip_patch_3d, params(), dll_dir=dll,
                                                          feed_direction=ms.FeedDirection.POS_X, boundary="MUR", theta_step_deg=5.0,
                                                          phi_step_deg=10.0, mesh_quality=5, work_dir=os.path.join(work, "c"))
    pitch = 0.0612
    arr = [PatchInstance(name=f"P{n}", params=params(), center_x_m=(ix - 0.5) * pitch, center_y_m=(iy - 0.5) * pitch, center_z_m=0.0,
                         feed_direction=ms.FeedDirection.NEG_X) for n, (ix, iy) in enumerate([(0, 0), (1, 0), (0, 1), (1, 1)])]
    scenes["multi_2x2"], prep_mm = capture(mm.prepare_openems_microstrip_multi_3d, arr, dll_dir=dll, boundary="PML_8", theta_step_deg=2.0,
                                           phi_step_deg=5.0, mesh_quality=3, work_dir=os.path.join(work, "d"))
    rot = [PatchInstance(name="R1", params=params(), center_x_m=0.0, center_y_m=0.0, center_z_m=0.01, feed_direction=ms.FeedDirection.NEG_Y,
                         rot_x_deg=0.0, rot_y_deg=0.0, rot_z_deg=90.0),
           PatchInstance(name="R2", params=params(), center_x_m=0.08, center_y_m=0.0, center_z_m=0.0, feed_direction=ms.FeedDirection.POS_X,
                         rot_x_deg=90.0, rot_y_deg=0.0, rot_z_deg=0.0)]
    scenes["multi_rotated"], _ = capture(mm.prepare_openems_microstrip_multi_3d, rot, dll_dir=dll, boundary="MUR", theta_step_deg=4.0,
                                         phi_step_deg=10.0, mesh_quality=6, nf_center_mode="centroid", end_criteria_db=-40.0,
                                         work_dir=os.path.join(work, "d"))
    scenes["multi_manual_box"], _ = capture(mm.prepare_openems_microstrip_multi_3d, arr[:1], dll_dir=dll, boundary="MUR", simbox_mode="manual",
                                            manual_size_mm=(260.0, 240.0, 200.0), mesh_quality=2, work_dir=os.path.join(work, "d"))
    scenes["legacy_2g45"], prep_lg = capture(lg.prepare_openems_patch, params(), dll_dir=dll, work_dir=os.path.join(work, "e"))
    json.dump(scenes, open(os.path.join(OUT, "scene_calls.json"), "w"))

    # ---- result conversion ------------------------------------------------------------------------
    conv = {}

    def run(fn, prep, f, sub=None):
        rec.calls = []
        r = fn(prep, frequency_hz=f, verbose=0)
        assert r.ok, r.message
        if sub is not None:      # a 91 x 181 grid: every sub[0]-th row and sub[1]-th column, the shape and two checksums
            a = np.asarray(r.intensity, float)
            return {"theta": _f(r.theta), "phi": _f(r.phi), "shape": list(a.shape), "sub": list(sub),
                    "intensity_sub": _f(a[::sub[0], ::sub[1]]), "sum": float(a.sum()), "sum_sq": float((a * a).sum()),
                    "is_dBi": bool(r.is_dBi), "message": r.message,
                    "calc_calls": [c for c in rec.calls if c["op"] == "CalcNF2FF"][:3],
                    "n_calc_calls": sum(1 for c in rec.calls if c["op"] == "CalcNF2FF")}
        return {"theta": _f(r.theta), "phi": _f(r.phi), "intensity": _f(r.intensity), "is_dBi": bool(r.is_dBi),
                "message": r.message, "calc_calls": [c for c in rec.calls if c["op"] == "CalcNF2FF"][:3],
                "n_calc_calls": sum(1 for c in rec.calls if c["op"] == "CalcNF2FF")}

    conv["fixed"] = run(fx.run_prepared_openems_fixed, prep_fixed, 2.45e9)
    conv["microstrip"] = run(ms.run_prepared_openems_microstrip, prep_ms, 2.45e9)
    conv["microstrip3d"] = run(m3.run_prepared_openems_microstrip_3d, prep_3d, 5.8e9)
    conv["multi"] = run(mm.run_prepared_openems_microstrip_multi_3d, prep_mm, 2.45e9)
    # legacy variant: radian theta/phi in Prepared (openems.py:262-263), degrees handed to CalcNF2FF (:299-300), result in
    # radians; three branches of the conversion (:330-408)
    for mode, key in (("full", "legacy"), ("weak", "legacy_enorm_fallback"), ("fields", "legacy_fields_only")):
        install_fake.NF2FF.mode = mode
        conv[key] = run(lg.run_prepared_openems, prep_lg, 2.45e9, sub=(5, 9))
    install_fake.NF2FF.mode = "enorm"
    json.dump(conv, open(os.path.join(OUT, "result_conversion.json"), "w"))

    # ---- input model ------------------------------------------------------------------------------
    p = params(metal="gold", metal_thickness_um=3.0)
    model = {"from_user_units": json.loads(p.model_dump_json()),
             "defaults": json.loads(PatchAntennaParams(frequency_hz=1e9, eps_r=2.2, h_m=1e-3).model_dump_json()),
             "props": {"frequency_ghz": p.frequency_ghz, "h_mm": p.h_mm, "L_mm": p.L_mm, "W_mm": p.W_mm},
             "feed_directions": {m.name: m.value for m in ms.FeedDirection}}
    bad = []
    for kw in ({"frequency_hz": -1, "eps_r": 2, "h_m": 1e-3}, {"frequency_hz": 1e9, "eps_r": 1.0, "h_m": 1e-3},
               {"frequency_hz": 1e9, "eps_r": 2, "h_m": 0}, {"frequency_hz": 1e9, "eps_r": 2, "h_m": 1e-3, "loss_tangent": -0.1}):
        try:
            PatchAntennaParams(**kw)
            bad.append(False)
        except Exception:
            bad.append(True)
    model["rejects"] = bad
    json.dump(model, open(os.path.join(OUT, "input_model.json"), "w"), indent=1)
    print("wrote", sorted(f for f in os.listdir(OUT) if f.endswith(".json")))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""What float32 costs over a whole run: the float32 oracle against its double-precision build (oracle/Makefile:
libfdtd_oracle_f64.so, -DFDTD_REAL=double — same algorithm, same C ABI, state and coefficients in double).

The HIP library equals the float32 oracle bit for bit (tests/test_parity_gpu.py), so this is also the HIP library's distance
from the exact solution of the same difference equations.  north_star asks for 1e-3 relative L2 on S11(f) and the E/H-plane
patterns against a CPU reference that itself steps in float32 ([EXT] openEMS, FDTD_FLOAT = float); the bar here is 1e-4, an
order inside it.  Two double runs per scene: "tables_f32" sees the float32-rounded tables of the C ABI (difference = rounding
of the time stepping alone), "tables_f64" the same tables in double (adds the rounding of the coefficients).

    python tests/fp32_error_budget.py [--scenes C1,NS] [--steps 12000] > profiles/r04/fp32_error_budget.json

CPU only (runs in the build container: C1 seconds, NS a few minutes per run on 8 cores).  Lives under tests/ because it
loads the oracle."""
import argparse
import ctypes
import importlib
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
PKG = "fdtd-solver-antennas_amd"
import helpers  # noqa: E402


def rel_l2(a, b):
    a, b = np.ravel(a), np.ravel(b)
    n = min(a.size, b.size)
    return float(np.linalg.norm(a[:n] - b[:n]) / np.linalg.norm(b[:n]))


def libs():
    capi = importlib.import_module(PKG + "._capi")
    subprocess = importlib.import_module("subprocess")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    f32 = capi.bind(ctypes.CDLL(os.path.join(ROOT, "oracle", "libfdtd_oracle.so")))
    f64 = helpers.load_oracle_f64()
    n = len(os.sched_getaffinity(0))
    f32.fdtd_oracle_set_threads(n); f64.fdtd_oracle_set_threads(n)
    return f32, f64, n


def compare(res, ref):
    """res / ref: dicts with u, i, s11 (complex), cuts (linear [ntheta][2]), Dmax, steps"""
    out = {"steps": [res["steps"], ref["steps"]],
           "port_u": rel_l2(res["u"], ref["u"]), "port_i": rel_l2(res["i"], ref["i"]),
           "s11": rel_l2(res["s11"], ref["s11"]),
           "e_plane": rel_l2(res["cuts"][:, 0], ref["cuts"][:, 0]), "h_plane": rel_l2(res["cuts"][:, 1], ref["cuts"][:, 1]),
           "Dmax": abs(res["Dmax"] - ref["Dmax"]) / ref["Dmax"]}
    out["max"] = max(v for k, v in out.items() if k != "steps")
    return out


def scene_C1(f32, f64):
    """The reference GUI's default scene through the plugin surface (prepare_*_patch_fixed: graded mesh, MUR, -40 dB end
    criterion: solver_fdtd_openems_fixed.py:113-342)."""
    sol = importlib.import_module(PKG + ".solver_fdtd_hip")
    par = importlib.import_module(PKG + ".params")
    simm = importlib.import_module(PKG + ".simulation")
    p = par.PatchAntennaParams.from_user_units(frequency_ghz=2.45, er=4.3, h_mm=1.6, loss_tangent=0.02)
    plain_build = simm.Simulation.build
    out = {}
    for tag, lib, dbl in (("f32", f32, False), ("f64_tables_f32", f64, False), ("f64_tables_f64", f64, True)):
        def build(self, l, **kw):
            if dbl:
                helpers.stage_f64_tables(self, l)
            return plain_build(self, l, **kw)
        simm.Simulation.build = build
        try:
            with tempfile.TemporaryDirectory() as td:
                t0 = time.perf_counter()
                prep = sol.prepare_hip_patch_fixed(p, work_dir=os.path.join(td, "w"), lib=lib)
                assert prep.ok, prep.message
                r = sol.run_prepared_hip(prep, frequency_hz=p.frequency_hz, verbose=0)
                assert r.ok, r.message
                out[tag] = {"u": r.port_u, "i": r.port_i, "s11": r.s11, "cuts": 10.0 ** (np.asarray(r.intensity) / 20.0),
                            "Dmax": r.Dmax, "steps": r.stats["steps"], "seconds": time.perf_counter() - t0,
                            "grid": r.stats["grid"], "energy_db": r.stats["energy_db"]}
        finally:
            simm.Simulation.build = plain_build
    return out


def scene_workload(name, steps, f32, f64):
    wl = importlib.import_module(PKG + ".workloads")
    sc = importlib.import_module(PKG + ".scene")
    simm = importlib.import_module(PKG + ".simulation")
    nf = importlib.import_module(PKG + ".nf2ff")
    oa = importlib.import_module(PKG + ".openems_api")
    w = wl.baseline_workload(name)
    vox = sc.voxelize(w.scene, w.grid)
    th = np.deg2rad(np.arange(0.0, 181.0, 2.0))
    ph = np.deg2rad(np.array([0.0, 90.0]))
    f = np.linspace(max(1e9, 0.7 * w.f0), 1.3 * w.f0, 201)
    out = {}
    for tag, lib, dbl in (("f32", f32, False), ("f64_tables_f32", f64, False), ("f64_tables_f64", f64, True)):
        s = simm.Simulation(w.grid, vox, f0=w.f0, fc=w.fc, boundary="CPML", cpml_cells=10, nr_ts=steps,
                            end_criteria=0.0, nf2ff_freqs=[w.f0])
        e = helpers.build_f64(s, lib, double_tables=True) if dbl else s.build(lib)
        t0 = time.perf_counter()
        e.run(steps)
        dt = time.perf_counter() - t0
        u, i = s.port_series()[0]
        t = np.arange(u.size) * s.dt
        U, I = oa.dft_time2freq(t, u, f), oa.dft_time2freq(t + 0.5 * s.dt, i, f)
        inc = 0.5 * (U + 50.0 * I)
        ff = nf.calc_nf2ff(lib, s.nf2ff_box, s.nf2ff_boxes(), [w.f0], th, ph, [0.0, 0.0, 1e-3])
        out[tag] = {"u": u, "i": i, "s11": (U - inc) / inc, "cuts": np.asarray(ff.E_norm[0]), "Dmax": float(ff.Dmax[0]),
                    "steps": steps, "seconds": dt, "grid": list(w.grid.shape), "energy_db": None}
        del e, s
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", default="C1,NS")
    ap.add_argument("--steps", type=int, default=12000)
    args = ap.parse_args()
    f32, f64, cores = libs()
    doc = {"what": "float32 oracle (== libfdtd_hip.so bit for bit) vs the double build of the same oracle; relative L2 (Dmax: relative difference)",
           "bar": 1e-4, "north_star_tolerance": 1e-3, "host_cores": cores, "scenes": {}}
    for name in args.scenes.split(","):
        r = scene_C1(f32, f64) if name == "C1" else scene_workload(name, args.steps, f32, f64)
        k = int(np.argmin(np.abs(r["f64_tables_f64"]["s11"])))
        entry = {"grid": r["f32"]["grid"], "steps_f32": r["f32"]["steps"],
                 "seconds": {t: round(v["seconds"], 2) for t, v in r.items()},
                 "s11_min_dB_f64": round(float(20 * np.log10(np.abs(r["f64_tables_f64"]["s11"][k]))), 4),
                 "Dmax_dBi": {t: round(float(10 * np.log10(v["Dmax"])), 6) for t, v in r.items()},
                 "energy_db": {t: v["energy_db"] for t, v in r.items()},
                 "f32_vs_f64_with_f32_tables (rounding of the time stepping)": compare(r["f32"], r["f64_tables_f32"]),
                 "f32_vs_f64_with_f64_tables (+ rounding of the coefficients)": compare(r["f32"], r["f64_tables_f64"])}
        entry["within_bar"] = entry["f32_vs_f64_with_f64_tables (+ rounding of the coefficients)"]["max"] <= doc["bar"]
        doc["scenes"][name] = entry
        print(f"[fp32 budget] {name}: {json.dumps(entry)}", file=sys.stderr, flush=True)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""North-star acceptance run: the 300x300x60 patch-on-FR-4 workload (CPML-10, lumped port, NF2FF surfaces)
time-stepped to completion on the MI355X library and on the CPU oracle; reports S11(f) and E/H-plane pattern
agreement (relative L2) plus both throughputs.  Writes one JSON object to stdout.

    python tests/acceptance_northstar.py [--steps 12000] [--workload NS]

Lives under tests/ because it loads the oracle (test infrastructure); `test_parity_gpu.py::
test_northstar_acceptance` runs a shorter version of it in the -m gpu suite.
"""
import argparse
import ctypes
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "fdtd-solver-antennas_amd"


def rel_l2(a, b):
    return float(np.linalg.norm(np.ravel(a) - np.ravel(b)) / np.linalg.norm(np.ravel(b)))


def run(steps=12000, workload="NS"):
    args = argparse.Namespace(steps=steps, workload=workload)
    capi = importlib.import_module(PKG + "._capi")
    wl = importlib.import_module(PKG + ".workloads")
    sc = importlib.import_module(PKG + ".scene")
    simm = importlib.import_module(PKG + ".simulation")
    nf = importlib.import_module(PKG + ".nf2ff")
    oa = importlib.import_module(PKG + ".openems_api")
    sys.path.insert(0, ROOT)
    import bench
    hip = capi.load_hip_library()
    ora = capi.bind(ctypes.CDLL(os.path.join(ROOT, "oracle", "libfdtd_oracle.so")))
    ora.fdtd_oracle_set_threads(bench.usable_cores())
    w = wl.baseline_workload(args.workload)
    vox = sc.voxelize(w.scene, w.grid)
    th = np.deg2rad(np.arange(0.0, 181.0, 2.0))
    ph = np.deg2rad(np.array([0.0, 90.0]))
    f = np.linspace(max(1e9, 0.7 * w.f0), 1.3 * w.f0, 201)
    out, res = {"workload": args.workload, "grid": list(w.grid.shape), "steps": args.steps}, {}
    for tag, lib in (("gpu", hip), ("cpu_oracle", ora)):
        s = simm.Simulation(w.grid, vox, f0=w.f0, fc=w.fc, boundary="CPML", cpml_cells=10, nr_ts=args.steps,
                            end_criteria=0.0, nf2ff_freqs=[w.f0])
        e = s.build(lib)
        t0 = time.perf_counter()
        e.run(args.steps)
        dt = time.perf_counter() - t0
        u, i = s.port_series()[0]
        t = np.arange(u.size) * s.dt
        U, I = oa.dft_time2freq(t, u, f), oa.dft_time2freq(t + 0.5 * s.dt, i, f)
        inc = 0.5 * (U + 50.0 * I)
        s11 = (U - inc) / inc
        ff = nf.calc_nf2ff(lib, s.nf2ff_box, s.nf2ff_boxes(), [w.f0], th, ph, [0.0, 0.0, 1e-3])
        res[tag] = (s11, ff.E_norm[0], ff.Dmax[0], u, i)
        k = int(np.argmin(np.abs(s11)))
        out[tag] = {"seconds": round(dt, 3), "mcells_per_s": round(w.grid.ncells * args.steps / dt / 1e6, 1),
                    "s11_min_dB": round(float(20 * np.log10(np.abs(s11[k]))), 3), "f_at_min_GHz": round(float(f[k] / 1e9), 4),
                    "Dmax_dBi": round(float(10 * np.log10(ff.Dmax[0])), 4), "backend": e.backend}
        del e, s
    g, c = res["gpu"], res["cpu_oracle"]
    out["rel_l2"] = {"s11": rel_l2(g[0], c[0]), "e_plane": rel_l2(g[1][:, 0], c[1][:, 0]), "h_plane": rel_l2(g[1][:, 1], c[1][:, 1]),
                     "port_u": rel_l2(g[3], c[3]), "port_i": rel_l2(g[4], c[4]), "Dmax": abs(g[2] - c[2]) / c[2]}
    out["tolerance"] = 1e-3
    out["pass"] = all(v < 1e-3 for v in out["rel_l2"].values())
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=12000)
    ap.add_argument("--workload", default="NS")
    args = ap.parse_args()
    print(json.dumps(run(args.steps, args.workload)))


if __name__ == "__main__":
    main()

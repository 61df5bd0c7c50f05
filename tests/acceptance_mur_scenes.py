#!/usr/bin/env python3
"""The reference's larger default scenes with MUR faces — the 2 x 2 multi-patch array (solver_fdtd_openems_microstrip_multi_3d.py:102) and the
3-D microstrip patch at 5.8 GHz (solver_fdtd_openems_microstrip_3d.py) — run to their end through the plugin surface on the MI355X library
(two launches per timestep, no Mur apply pass) and on the CPU oracle: timesteps, end energy, every port's series and S11, the pattern, Dmax.

    python tests/acceptance_mur_scenes.py          # on a GPU box; a few tens of seconds of oracle time

Lives under tests/ because it loads the oracle (test infrastructure).
"""
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
PKG = "fdtd-solver-antennas_amd"


def rel_l2(a, b):
    return float(np.linalg.norm(np.ravel(a) - np.ravel(b)) / max(np.linalg.norm(np.ravel(b)), 1e-300))


def main():
    capi = importlib.import_module(PKG + "._capi")
    s = importlib.import_module(PKG + ".solver_fdtd_hip")
    P = importlib.import_module(PKG + ".params").PatchAntennaParams
    import bench
    hip = capi.load_hip_library()
    ora = capi.bind(ctypes.CDLL(os.path.join(ROOT, "oracle", "libfdtd_oracle.so")))
    ora.fdtd_oracle_set_threads(bench.usable_cores())
    p245 = P.from_user_units(frequency_ghz=2.45, er=4.3, h_mm=1.6, loss_tangent=0.02)
    p58 = P.from_user_units(frequency_ghz=5.8, er=4.3, h_mm=1.6, loss_tangent=0.02)
    tmp = tempfile.mkdtemp()
    scenes = [
        ("multi_3d 2x2 2.45 GHz MUR", 2.45e9, lambda lib, d: s.prepare_hip_microstrip_multi_3d(
            [s.PatchInstance(f"P{n}", p245, (ix - 0.5) * 0.0612, (iy - 0.5) * 0.0612, 0.0, s.FeedDirection.NEG_X)
             for n, (ix, iy) in enumerate([(0, 0), (1, 0), (0, 1), (1, 1)])], work_dir=d, lib=lib)),
        ("microstrip_3d 5.8 GHz MUR", 5.8e9, lambda lib, d: s.prepare_hip_microstrip_patch_3d(p58, work_dir=d, lib=lib)),
    ]
    for name, f0, prep_fn in scenes:
        out = []
        for tag, lib in (("gpu", hip), ("cpu_oracle", ora)):
            prep = prep_fn(lib, os.path.join(tmp, name.split()[0] + tag))
            assert prep.ok, prep.message
            t0 = time.perf_counter()
            r = s.run_prepared_hip(prep, frequency_hz=f0, verbose=0)
            dt = time.perf_counter() - t0
            assert r.ok, r.message
            ports = prep.ports or [prep.port]
            series = prep.FDTD.sim.port_series()
            s11 = [s.s11_from_port(port, prep.sim_path, f0)[1] for port in ports]
            info = prep.FDTD.sim.engine.schedule_info() if tag == "gpu" else None
            out.append((r, series, s11, dt, info))
        (g, sg, s11g, tg, info), (c, sc_, s11c, tc, _) = out
        rec = {"scene": name, "grid": g.stats["grid"], "timesteps": [g.stats["steps"], c.stats["steps"]], "end_energy_db": [round(g.stats["energy_db"], 4), round(c.stats["energy_db"], 4)],
               "launches_per_timestep_gpu": info["launches_per_timestep"], "gcells_per_s_gpu": round(g.stats["mcells_per_s"] / 1e3, 1), "gcells_per_s_oracle": round(c.stats["mcells_per_s"] / 1e3, 2),
               "call_seconds": [round(tg, 2), round(tc, 2)],
               "port_u_rel_l2": [rel_l2(a[0], b[0]) for a, b in zip(sg, sc_)], "port_i_rel_l2": [rel_l2(a[1], b[1]) for a, b in zip(sg, sc_)],
               "s11_rel_l2": [rel_l2(a, b) for a, b in zip(s11g, s11c)], "intensity_rel_l2": rel_l2(g.intensity, c.intensity),
               "Dmax_dBi": [round(10 * np.log10(g.Dmax), 4), round(10 * np.log10(c.Dmax), 4)]}
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()

"""Small grids inside the multi-timestep launches: us per timestep for a list of patch-scene shapes, (a) under a list of
occupancy caps ($FDTD_OCC_WF = blocks per CU; 0 = the launcher's automatic cap), (b) with one launch per timestep
($FDTD_WF_MULTI=1) for comparison.  profiles/r03/occupancy_cap_sweep_multi_timestep_launches.txt and
small_grids_multi_timestep_vs_single.txt came from this script.

    python tools/small_grid_schedules.py [caps, e.g. 0,6,5,4,3,2,1] [steps]        # on a GPU box
"""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = "fdtd-solver-antennas_amd"
capi = importlib.import_module(PKG + "._capi"); wl = importlib.import_module(PKG + ".workloads")
sc = importlib.import_module(PKG + ".scene"); simm = importlib.import_module(PKG + ".simulation")
hip = capi.load_hip_library()
caps = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0").split(",")]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
shapes = [(56, 55, 50), (64, 60, 36), (100, 100, 40), (128, 128, 40), (150, 150, 40), (180, 180, 40), (200, 200, 40), (220, 220, 40),
          (200, 200, 48), (300, 300, 60)]
for shp in shapes:
    w = wl.patch_workload("t", nx=shp[0], ny=shp[1], nz=shp[2]); vox = sc.voxelize(w.scene, w.grid)
    row, info = [], None
    for tag in [f"cap{c}" if c else "auto" for c in caps] + ["one-launch-per-timestep"]:
        os.environ.pop("FDTD_OCC_WF", None)
        os.environ["FDTD_WF_MULTI"] = "1" if tag == "one-launch-per-timestep" else "64"
        if tag.startswith("cap"):
            os.environ["FDTD_OCC_WF"] = tag[3:]
        sim = simm.Simulation(w.grid, vox, f0=w.f0, fc=w.fc, boundary="CPML", cpml_cells=10, nr_ts=3 * steps + 2000, nf2ff_freqs=[w.f0])
        e = sim.build(hip)
        e.run(1000)
        for _ in range(6):
            e.run(64)
        t0 = time.perf_counter(); e.run(steps); dt = time.perf_counter() - t0
        info = e.schedule_info()
        row.append((tag, round(dt / steps * 1e6, 2)))
        del e, sim
    print(shp, "blocks per half-step", info["blocks_per_sweep"], row, flush=True)

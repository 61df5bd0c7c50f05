"""Time-to-first-step of the BASELINE workloads: scene -> voxels -> operator -> engine ready, device operator build
(fdtd_build_operator, product default) against the host numpy build + upload."""
import os, sys, time, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = "fdtd-solver-antennas_amd"
capi = importlib.import_module(PKG + "._capi"); wl = importlib.import_module(PKG + ".workloads")
sc = importlib.import_module(PKG + ".scene"); simm = importlib.import_module(PKG + ".simulation")
hip = capi.load_hip_library()
names = sys.argv[1:] or ["NS", "C3"]
for name in names:
    t0 = time.perf_counter(); w = wl.baseline_workload(name); vox = sc.voxelize(w.scene, w.grid); t1 = time.perf_counter()
    for dev in (True, False, True):
        ta = time.perf_counter()
        sim = simm.Simulation(w.grid, vox, f0=w.f0, fc=w.fc, boundary="CPML", cpml_cells=10, nr_ts=1000, nf2ff_freqs=[w.f0], device_operator=dev)
        e = sim.build(hip)
        e.run(1)
        tb = time.perf_counter()
        print(f"{name} ({w.grid.ncells/1e6:.1f} Mcells): scene+voxelize {t1-t0:.2f} s; Simulation()+build()+first step, "
              f"{'device' if dev else 'host numpy'} operator build: {tb-ta:.2f} s  form {e.operator_form()}", flush=True)
        del e, sim

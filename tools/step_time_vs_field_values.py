"""Does the step time depend on the field VALUES?  NS / C3, PEC and CPML: source-driven fields (as kernel_ab), all-zero fields
(no steps before timing... the source still runs), dense random fields."""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
PKG = "fdtd-solver-antennas_amd"
capi = importlib.import_module(PKG + "._capi"); wl = importlib.import_module(PKG + ".workloads")
sc = importlib.import_module(PKG + ".scene"); simm = importlib.import_module(PKG + ".simulation")
lib = capi.load_hip_library()
steps = 500
for name in sys.argv[1].split(","):
    w = wl.baseline_workload(name); vox = sc.voxelize(w.scene, w.grid)
    for bc in ("PEC", "CPML"):
        for fill in ("source", "random", "random_small"):
            sim = simm.Simulation(w.grid, vox, f0=w.f0, fc=w.fc, boundary=bc, cpml_cells=10, nr_ts=8 * steps, nf2ff_freqs=[w.f0])
            eng = sim.build(lib, flags=1)
            if fill != "source":
                rng = np.random.default_rng(1)
                for kind in (0, 1):
                    for c in range(3):
                        eng.set_field(kind, c, ((1e-3 if fill == "random" else 1e-20) * rng.standard_normal(eng.local_shape)).astype(np.float32))
            out = []
            for rep in range(4):
                t0 = time.perf_counter(); eng.run(steps); dt = time.perf_counter() - t0
                out.append(round(dt / steps * 1e6, 1))
            f = eng.get_field(0, 2)
            print(json.dumps({"workload": name, "bc": bc, "fill": fill, "us_step_by_500_steps": out,
                              "nonzero_frac": float(np.count_nonzero(f)) / f.size, "max_abs": float(np.abs(f).max())}), flush=True)
            del eng, sim

"""Long-run check: the AUTO schedule (several timesteps per launch on the cache-resident NS / C2, one launch per timestep with H behind
E on C3 / C5) vs two launches per timestep, full size, from step 0: fields and port series bit for bit.  [names:steps,...] optional."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = "fdtd-solver-antennas_amd"
capi = importlib.import_module(PKG + "._capi"); wl = importlib.import_module(PKG + ".workloads")
sc = importlib.import_module(PKG + ".scene"); simm = importlib.import_module(PKG + ".simulation")
lib = capi.load_hip_library()
cases = [(c.split(":")[0], int(c.split(":")[1])) for c in sys.argv[1].split(",")] if len(sys.argv) > 1 else [("NS", 12000), ("C2", 10000), ("C3", 6000), ("C5", 1500)]
for name, steps in cases:
    w = wl.baseline_workload(name); vox = sc.voxelize(w.scene, w.grid)
    out = []
    for flags in (1, 0):
        sim = simm.Simulation(w.grid, vox, f0=w.f0, fc=w.fc, boundary="CPML", cpml_cells=10, nr_ts=steps + 8, nf2ff_freqs=[w.f0])
        e = sim.build(lib, flags=flags)
        t0 = time.perf_counter()
        for n in (steps // 3, steps // 3, steps - 2 * (steps // 3)):
            e.run(n)
        dt = time.perf_counter() - t0
        f = [e.get_field(k, c) for k in (0, 1) for c in range(3)]
        out.append((f, sim.port_series()[0], dt))
        del e, sim
    same = all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(out[0][0], out[1][0]))
    u = np.array_equal(out[0][1][0], out[1][1][0]) and np.array_equal(out[0][1][1], out[1][1][1])
    print(f"{name}: {steps} steps, two launches per timestep {out[0][2]:.2f} s, AUTO {out[1][2]:.2f} s; fields identical bit for bit: {same}; port series identical: {u}; max|V| {max(float(np.abs(a).max()) for a in out[0][0][:3]):.3e}", flush=True)

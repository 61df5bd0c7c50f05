"""Per-slab step time of a z-slab decomposition, each slab timed ALONE on one GPU: max / mean over the ranks for the
even split and for the cost-weighted one (simulation.slab_partition).

Every slab runs as it would in the N-rank job — same planes, same CPML layers, same kernels, the p2p mailbox protocol in
its update kernels — with its halos going to ITSELF (FDTD_FLAG_LOOPBACK: include/fdtd_hip.h).  An end slab thereby pushes
one halo plane more than in a real run: its time is an upper bound.  Fields: seeded noise (the step time depends on the
field values; all-zero fields stream 6-15 % faster), no claim about their physics.

    python tools/slab_balance.py [NS:8,C4:4,C5:8] [steps]        # on a GPU box; prints a table per configuration
"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "fdtd-solver-antennas_amd"


def main():
    cfgs = [(c.split(":")[0], int(c.split(":")[1])) for c in (sys.argv[1] if len(sys.argv) > 1 else "NS:8,C4:4,C5:8").split(",")]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    capi = importlib.import_module(PKG + "._capi")
    wl = importlib.import_module(PKG + ".workloads")
    sc = importlib.import_module(PKG + ".scene")
    simm = importlib.import_module(PKG + ".simulation")
    hip = capi.load_hip_library()
    print(f"# tools/slab_balance.py: us per timestep of every slab, timed alone (halos to itself), {steps} timesteps, "
          f"z-layer plane cost {simm.Z_LAYER_PLANE_COST}", flush=True)
    for name, world in cfgs:
        w = wl.baseline_workload(name)
        vox = sc.voxelize(w.scene, w.grid)
        nz = w.grid.shape[2]
        rows = {}
        for part in ("even", "cost"):
            for rank in range(world):
                sim = simm.Simulation(w.grid, vox, f0=w.f0, fc=w.fc, boundary="CPML", cpml_cells=10, nr_ts=steps + 400,
                                      nf2ff_freqs=[w.f0])
                e = sim.build(hip, rank=rank, world=world, flags=capi.FLAG_LOOPBACK, partition=part)
                blob = e.p2p_export()
                e.p2p_attach(blob, blob)
                rng = np.random.default_rng(rank)
                for kind in (0, 1):
                    for c in range(3):
                        e.set_field(kind, c, (1e-3 * rng.standard_normal(e.local_shape)).astype(np.float32))
                e.run(300)
                t0 = time.perf_counter()
                e.run(steps)
                dt = time.perf_counter() - t0
                zl = simm.plane_costs(nz, 10, 10, w_layer=2.0)[e.k0:e.k0 + e.nk] > 1.5
                vmax = float(np.nanmax(np.abs(e.get_field(0, 2))))
                rows[(part, rank)] = (e.nk, int(zl.sum()), dt / steps * 1e6, e.schedule_info()["launches_per_timestep"], vmax)
                del e, sim
        print(f"\n{name} {w.grid.shape[0]}x{w.grid.shape[1]}x{nz} over {world} ranks", flush=True)
        print("rank | even: planes (z-layer) launches us/step | cost: planes (z-layer) launches us/step")
        for rank in range(world):
            a, b = rows[("even", rank)], rows[("cost", rank)]
            print(f"{rank:4d} | {a[0]:4d} ({a[1]:2d}) {a[3]} {a[2]:9.2f}           | {b[0]:4d} ({b[1]:2d}) {b[3]} {b[2]:9.2f}")
        for part in ("even", "cost"):
            t = np.array([rows[(part, r)][2] for r in range(world)])
            fin = all(np.isfinite(rows[(part, r)][4]) for r in range(world))
            print(f"{part}: max {t.max():.2f} us, mean {t.mean():.2f} us, max/mean {t.max() / t.mean():.3f} -> "
                  f"{w.grid.ncells / t.max() / 1e3:.1f} Gcells/s if every rank waits for the slowest (fields finite: {fin})", flush=True)


if __name__ == "__main__":
    main()

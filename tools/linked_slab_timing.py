"""Per-step cost of N linked z-slabs driven by one process on ONE GPU (fdtd_link / fdtd_run_linked): the excess over
world = 1 is the host + launch overhead of the multi-slab schedule."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from importlib import import_module as im
capi = im("fdtd-solver-antennas_amd._capi"); wl = im("fdtd-solver-antennas_amd.workloads"); sc = im("fdtd-solver-antennas_amd.scene"); simm = im("fdtd-solver-antennas_amd.simulation")
import torch
hip = capi.load_hip_library()
w = wl.baseline_workload("NS"); vox = sc.voxelize(w.scene, w.grid)
for world, flag in ((1, 0), (2, 0x20), (2, 0x40), (8, 0x20), (8, 0x40)):
    sims = [simm.Simulation(w.grid, vox, f0=w.f0, fc=w.fc, boundary="CPML", cpml_cells=10, nr_ts=3000, nf2ff_freqs=[w.f0]) for _ in range(world)]
    engs = [s.build(hip, rank=r, world=world, flags=flag) for r, s in enumerate(sims)]
    run = (lambda n: engs[0].run(n)) if world == 1 else (lambda n: capi.run_linked(engs, n))
    run(100)
    t0 = time.perf_counter(); run(500); dt = time.perf_counter() - t0
    print(f"world {world} flag {flag:#x}: {dt/500*1e6:.1f} us per step on ONE GPU ({w.grid.ncells*500/dt/1e6:.0f} Mcells/s); per-slab host+launch cost visible as the excess over world 1", flush=True)
    del engs, sims

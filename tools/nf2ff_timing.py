"""Row a10 on the big configurations: what CalcNF2FF costs after a run — the device transform of the 24 recorded face boxes
to one frequency (fdtd_rec_transform, incl. the copies to the host), the host-side node interpolation / equivalent currents
(nf2ff.NF2FFBox.surface_currents) and the radiation integral on the GPU (fdtd_farfield) for the 3-D variants' 91 x 73
directions (the reference loops CalcNF2FF over 73 phi values: solver_fdtd_openems_microstrip_3d.py:224-238).

    python tools/nf2ff_timing.py [C4,C5] [steps]        # on a GPU box
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
    names = (sys.argv[1] if len(sys.argv) > 1 else "C4,C5").split(",")
    capi = importlib.import_module(PKG + "._capi")
    wl = importlib.import_module(PKG + ".workloads")
    sc = importlib.import_module(PKG + ".scene")
    simm = importlib.import_module(PKG + ".simulation")
    nf = importlib.import_module(PKG + ".nf2ff")
    hip = capi.load_hip_library()
    th = np.deg2rad(np.arange(0.0, 181.0, 2.0))
    ph = np.deg2rad(np.arange(0.0, 361.0, 5.0))
    print(f"# tools/nf2ff_timing.py: CalcNF2FF for {th.size} x {ph.size} directions at one frequency after a run; seconds", flush=True)
    for name in names:
        w = wl.baseline_workload(name)
        steps = int(sys.argv[2]) if len(sys.argv) > 2 else w.steps
        vox = sc.voxelize(w.scene, w.grid)
        sim = simm.Simulation(w.grid, vox, f0=w.f0, fc=w.fc, boundary="CPML", cpml_cells=10, nr_ts=steps, nf2ff_freqs=[w.f0],
                              nf2ff_mode="auto")
        e = sim.build(hip)
        t0 = time.perf_counter(); e.run(steps); t_run = time.perf_counter() - t0
        t0 = time.perf_counter(); boxes = sim.nf2ff_boxes(); t_tr = time.perf_counter() - t0
        t0 = time.perf_counter(); boxes = sim.nf2ff_boxes(); t_tr2 = time.perf_counter() - t0
        centre = [0.0, 0.0, 1e-3]
        t0 = time.perf_counter(); pos, Js, Ms, flux = sim.nf2ff_box.surface_currents(boxes, 0, centre); t_sc = time.perf_counter() - t0
        TH, PH = np.meshgrid(th, ph, indexing="ij")
        k = 2 * np.pi * w.f0 / 299792458.0
        t0 = time.perf_counter(); capi.farfield(hip, pos, Js, Ms, k, TH.ravel(), PH.ravel()); t_ff = time.perf_counter() - t0
        t0 = time.perf_counter(); res = nf.calc_nf2ff(hip, sim.nf2ff_box, boxes, [w.f0], th, ph, centre); t_all = time.perf_counter() - t0
        npts_box = sum(int(np.prod([r.hi[a] - r.lo[a] + 1 for a in range(3)])) for r in sim.nf2ff_box.requests)
        print(f"{name} {w.grid.shape}: {steps} timesteps in {t_run:.2f} s ({sim.nf2ff_mode}, {sim.dft_nsamples} samples of {npts_box} box points, "
              f"{sim.rec_bytes / 2**30:.2f} GiB); transform of the 24 boxes {t_tr:.3f} (again: {t_tr2:.3f}); surface currents on the host "
              f"{t_sc:.3f} ({pos.shape[0]} quadrature points); fdtd_farfield {t_ff:.3f}; calc_nf2ff (currents + integral) {t_all:.3f}; "
              f"Dmax {10 * np.log10(res.Dmax[0]):.2f} dBi", flush=True)
        del e, sim


if __name__ == "__main__":
    main()

"""Same-box A/B of the update kernels: one line per (workload, boundary) with the HIP-event durations of the two main
launches (raw, as measured) and the whole-step rate.  Variants of the library are compared by pointing
$FDTD_HIP_LIB_DIR at another build, variants of a run by environment switches the library reads at fdtd_create.

    python tools/kernel_ab.py NS,C3 [CPML,PEC] [steps]
"""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "fdtd-solver-antennas_amd"


def main():
    names = (sys.argv[1] if len(sys.argv) > 1 else "NS").split(",")
    bcs = (sys.argv[2] if len(sys.argv) > 2 else "CPML").split(",")
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 600
    capi = importlib.import_module(PKG + "._capi")
    wl = importlib.import_module(PKG + ".workloads")
    sc = importlib.import_module(PKG + ".scene")
    simm = importlib.import_module(PKG + ".simulation")
    libdir = os.environ.get("FDTD_HIP_LIB_DIR") or None
    if libdir:   # an older build (e.g. the previous round's kernels) may lack newer entry points this tool never calls
        import ctypes
        raw = ctypes.CDLL(capi.hip_library_path(libdir))
        for name in capi.ABI_SYMBOLS:
            try:
                getattr(raw, name)
            except AttributeError:
                setattr(raw, name, raw["fdtd_version"])
        lib = capi.bind(raw)
        capi._hip_lib = lib                      # the workload builders (C4, C5 go through solver_fdtd_hip) load "the" library
        os.environ.pop("FDTD_HIP_LIB_DIR", None)
    else:
        lib = capi.load_hip_library()
    for name in names:
        w = wl.baseline_workload(name)
        vox = sc.voxelize(w.scene, w.grid)
        for bc in bcs:
            bspec = bc
            if bc[0] in "xyz" and bc[1:] == "CPML":          # CPML on the two faces of one axis only, PEC elsewhere
                bspec = ["PEC"] * 6
                a = "xyz".index(bc[0])
                bspec[2 * a] = bspec[2 * a + 1] = "CPML"
            sim = simm.Simulation(w.grid, vox, f0=w.f0, fc=w.fc, boundary=bspec, cpml_cells=10, nr_ts=4 * steps + 2 * max(w.grid.shape) + 64,
                                  nf2ff_freqs=[w.f0])
            eng = sim.build(lib, flags=int(os.environ.get("AB_FLAGS", "0")))
            # nothing is timed before the pulse has reached every corner of the grid: the step time depends on the field values
            # (all-zero fields stream 6-15 % faster, profiles/r02/step_time_vs_field_values.txt)
            eng.run(max(steps // 2, 2 * max(w.grid.shape)))
            for _ in range(int(os.environ.get("AB_WARM_CALLS", "8"))):     # the XCD shares settle over the first fdtd_run calls of a context
                eng.run(64)
            t0 = time.perf_counter()
            eng.run(steps)
            dt = time.perf_counter() - t0
            prof = eng.run_profiled(min(steps, 2000))
            ov = getattr(prof, "ms_event_overhead", 0.0)
            print(json.dumps({"workload": name, "bc": bc, "tag": os.environ.get("AB_TAG", ""),
                              "gcells_s": round(w.grid.ncells * steps / dt / 1e9, 2),
                              "us_step": round(dt / steps * 1e6, 2),
                              "us_E_raw": round((prof.ms_update_e + ov) * 1e3, 2),
                              "us_H_raw": round((prof.ms_update_h + ov) * 1e3, 2)}), flush=True)
            del eng, sim


if __name__ == "__main__":
    main()

import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from importlib import import_module as im
capi = im("fdtd-solver-antennas_amd._capi"); wl = im("fdtd-solver-antennas_amd.workloads"); sc = im("fdtd-solver-antennas_amd.scene"); simm = im("fdtd-solver-antennas_amd.simulation")
hip = capi.load_hip_library()
def run(tag, w, vox, use_classes=True, no_src=False):
    if no_src:
        for p in vox.ports: p.port.excite = 0.0
    s = simm.Simulation(w.grid, vox, f0=w.f0, fc=w.fc, boundary="CPML", cpml_cells=10, nr_ts=400, nf2ff_freqs=[w.f0], use_classes=use_classes)
    e = s.build(hip); e.run(20)
    pr = e.run_profiled(40)
    print(tag, w.grid.shape, "op", s.operator_form, "nsrc", sum(len(p.src_idx) for p in vox.ports), "E %.1f us H %.1f us" % (1e3*pr.ms_update_e, 1e3*pr.ms_update_h), flush=True)
    for p in vox.ports: p.port.excite = 1.0
w = wl.baseline_workload("C5"); vox = sc.voxelize(w.scene, w.grid)
run("C5 multi scene", w, vox)
run("C5 multi scene, no sources", w, vox, no_src=True)
w2 = wl.patch_workload("C5"); vox2 = sc.voxelize(w2.scene, w2.grid)
run("C5 fixed scene", w2, vox2)
run("C5 multi scene raw op", w, vox, use_classes=False)

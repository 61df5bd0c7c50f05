"""One interior north-star slab (15 / 8 planes = rank 1 of 4 / 8) whose halos go to itself through the P2P mailbox transport:
us per step; FDTD_WAVEFRONT=0/1 and FDTD_WF_LAG select the schedule (profiles/r02/thin_slab_one_launch_vs_two.txt)."""
import os, sys, time, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = "fdtd-solver-antennas_amd"
capi = importlib.import_module(PKG + "._capi"); wl = importlib.import_module(PKG + ".workloads")
sc = importlib.import_module(PKG + ".scene"); simm = importlib.import_module(PKG + ".simulation")
hip = capi.load_hip_library()
w = wl.baseline_workload("NS"); vox = sc.voxelize(w.scene, w.grid)
for world in tuple(int(x) for x in os.environ.get("WORLDS", "4,8").split(",")):
    sim = simm.Simulation(w.grid, vox, f0=w.f0, fc=w.fc, boundary="CPML", cpml_cells=10, nr_ts=9000, nf2ff_freqs=[w.f0])
    e = sim.build(hip, rank=1, world=world)
    blob = e.p2p_export(); e.p2p_attach(blob, blob)
    e.run(300)
    t0 = time.perf_counter(); e.run(3000); dt = time.perf_counter() - t0
    print(f"dep_first={os.environ.get('FDTD_P2P_DEP_FIRST','0')} world {world} ({e.nk} planes): {dt/3000*1e6:.1f} us per step -> {w.grid.ncells*3000/dt/1e9:.1f} Gcells/s projected", flush=True)
    del e, sim

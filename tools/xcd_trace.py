"""Per-XCD timing of ONE k_step launch (diagnostic build of the library with -DFDTD_XCD_TRACE: tools/build_variant.sh trace
-DFDTD_XCD_TRACE): for every XCD (by HW_REG_XCC_ID) the first start and last end of its E blocks and of its H blocks, relative
to the launch's first block, and how many blocks it ran — what the cost-weighted XCD shares are supposed to equalise.

    FDTD_HIP_LIB_DIR=scratch/v/trace python tools/xcd_trace.py NS,C2 [FDTD_XCD_BALANCE values: 0,1]
"""
import ctypes, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "fdtd-solver-antennas_amd"
capi = importlib.import_module(PKG + "._capi"); wl = importlib.import_module(PKG + ".workloads")
sc = importlib.import_module(PKG + ".scene"); simm = importlib.import_module(PKG + ".simulation")
lib = capi.load_hip_library()
trace = lib.fdtd_debug_xcd_trace
trace.argtypes = [ctypes.c_longlong, ctypes.c_void_p]
NMAX = 65536
names = (sys.argv[1] if len(sys.argv) > 1 else "NS").split(",")
for name in names:
    w = wl.baseline_workload(name); vox = sc.voxelize(w.scene, w.grid)
    for bal in (sys.argv[2] if len(sys.argv) > 2 else "0,1").split(","):
        os.environ["FDTD_XCD_BALANCE"] = bal
        sim = simm.Simulation(w.grid, vox, f0=w.f0, fc=w.fc, boundary="CPML", cpml_cells=10, nr_ts=3000, nf2ff_freqs=[w.f0])
        e = sim.build(lib)
        trace(-1, None)
        e.run(1500)
        t_wall = __import__("time").perf_counter()
        nrun = 1000
        e.run(nrun)                              # back to back, no host synchronisation in between
        t_wall = (__import__("time").perf_counter() - t_wall) / nrun * 1e6
        last = int(e.step) - 1
        ends, lifes, cnts, launch = np.zeros((8, 2)), np.zeros((8, 2)), np.zeros((8, 2)), []
        for st in (last - 1, last):
            buf = np.zeros(NMAX * 4, np.uint64)
            trace(st, buf.ctypes.data_as(ctypes.c_void_p))
            r = buf.reshape(NMAX, 4)
            r = r[((r[:, 2] & 16) != 0) & (r[:, 3] == st)]
            t0 = r[:, 0].min()
            xid, role = (r[:, 2] & 7).astype(int), ((r[:, 2] >> 3) & 1).astype(int)
            for x in range(8):
                for ro in range(2):
                    m = (xid == x) & (role == ro)
                    if m.any():
                        ends[x, ro] += (r[m, 1].max() - t0) / 100.0 / 2
                        lifes[x, ro] += float((r[m, 1] - r[m, 0]).mean()) / 100.0 / 2
                        cnts[x, ro] += m.sum() / 2
            launch.append((r[:, 1].max() - t0) / 100.0)
        print(f"{name} FDTD_XCD_BALANCE={bal}: {t_wall:.2f} us per timestep over {nrun} back-to-back launches; the last two launches, us from the "
              f"launch's first block (XCD by XCC_ID: E blocks, last end, mean block lifetime | H blocks ...)")
        for x in range(8):
            print(f"  XCD {x}: E {cnts[x,0]:7.1f} end {ends[x,0]:7.2f} life {lifes[x,0]:6.2f}   H {cnts[x,1]:7.1f} end {ends[x,1]:7.2f} life {lifes[x,1]:6.2f}")
        print(f"  E ends: spread {ends[:,0].max() - ends[:,0].min():.2f} us; H ends: spread {ends[:,1].max() - ends[:,1].min():.2f} us; "
              f"launch {np.mean(launch):.2f} us, mean of the XCDs' last ends {ends[:,1].mean():.2f} us", flush=True)
        del e, sim

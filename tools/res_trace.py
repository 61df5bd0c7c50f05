"""Where a timestep of the resident schedule goes (csrc/resident.hip): per workgroup, wall-clock ticks (100 MHz) summed over the timesteps of ONE
launch, split into [E: wait for halos] [E: update, sources, Mur, publish] [H: wait] [H: update, publish].  Needs the diagnostic build:
    tools/build_variant.sh restrace -DFDTD_RES_TRACE ;  FDTD_HIP_LIB_DIR=$PWD/scratch/v/restrace python tools/res_trace.py [nx ny nz] [MUR|PEC]"""
import ctypes, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
PKG = "fdtd-solver-antennas_amd"
capi = importlib.import_module(PKG + "._capi")
from helpers import patch_sim
shape = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (56, 55, 50)
bc = sys.argv[4] if len(sys.argv) > 4 else "MUR"
raw = ctypes.CDLL(capi.hip_library_path())
hip = capi.load_hip_library()
s = patch_sim(*shape, boundary=bc, nr_ts=2000, nf2ff=False)
e = s.build(hip, flags=capi.FLAG_KERNEL_RESIDENT)
e.run(600)            # the pulse is in the grid
e.run(200)            # the traced launch (one launch: 200 <= 256 timesteps)
tab = np.zeros(1024 * 8, np.uint64)
assert raw.fdtd_debug_res_trace(tab.ctypes.data_as(ctypes.c_void_p)) == 0
tab = tab.reshape(1024, 8)[: e.schedule_info()["blocks_per_sweep"]].astype(float)
n = tab[:, 6].max()
ew, ec, hw, hc, tot, setup = (tab[:, q] / 100.0 for q in range(6))     # us
print(f"grid {shape} {bc}: {tab.shape[0]} workgroups, {int(n)} timesteps in the launch; us per timestep (mean over workgroups, [min .. max])")
for name, v in (("E wait", ew / n), ("E rest", (ec - ew) / n), ("H wait", hw / n), ("H rest", (hc - hw) / n), ("timestep", (ec + hc) / n)):
    print(f"  {name:9s} {v.mean():6.3f}  [{v.min():6.3f} .. {v.max():6.3f}]")
print(f"  launch total {tot.mean():8.1f} us, set-up {setup.mean():6.2f} us [{setup.min():.2f} .. {setup.max():.2f}]")

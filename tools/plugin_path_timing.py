"""Wall-clock of the whole plugin call (prepare_hip_patch_fixed + run_prepared_hip) for the reference's default
input (2.45 GHz patch on FR-4) on the MI355X backend, with a cProfile breakdown of the host side."""
import os, sys, time, importlib, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = "fdtd-solver-antennas_amd"
sol = importlib.import_module(PKG + ".solver_fdtd_hip"); par = importlib.import_module(PKG + ".params")
if os.environ.get("FDTD_HIP_LIB_DIR"):   # an older build of the library (same-box A/B): bind what it has
    import ctypes
    capi = importlib.import_module(PKG + "._capi")
    raw = ctypes.CDLL(capi.hip_library_path(os.environ["FDTD_HIP_LIB_DIR"]))
    for name in capi.ABI_SYMBOLS:
        if not hasattr(raw, name):
            setattr(raw, name, raw["fdtd_version"])
    capi._hip_lib = capi.bind(raw)
    os.environ.pop("FDTD_HIP_LIB_DIR")
import tempfile
params = par.PatchAntennaParams.from_user_units(frequency_ghz=2.45, er=4.3, h_mm=1.6, loss_tangent=0.02)
for rep in range(2):
    with tempfile.TemporaryDirectory() as td:
        pr = cProfile.Profile()
        t0 = time.perf_counter()
        pr.enable()
        prep = sol.prepare_hip_patch_fixed(params, work_dir=os.path.join(td, "w"))
        assert prep.ok, prep.message
        t1 = time.perf_counter()
        res = sol.run_prepared_hip(prep, frequency_hz=params.frequency_hz, verbose=0)
        pr.disable()
        t2 = time.perf_counter()
        assert res.ok, res.message
        st = res.stats if hasattr(res, "stats") else None
        print(f"rep {rep}: prepare {t1-t0:.3f} s, run_prepared {t2-t1:.3f} s; stats {st}; Dmax {getattr(res, 'Dmax', None)}", flush=True)
        if rep == 1:
            s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22); print(s.getvalue()[:6000])

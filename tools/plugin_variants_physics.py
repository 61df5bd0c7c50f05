"""Every plugin variant at the reference's default settings on the GPU: resonance, S11, Dmax, radiated / accepted power at the S11 minimum,
warnings (physics sanity, not parity).  python tools/plugin_variants_physics.py   # on a GPU box"""
import importlib, sys, time, tempfile, os, warnings
import numpy as np
sys.path.insert(0, "/root/repo")
PKG = "fdtd-solver-antennas_amd"
s = importlib.import_module(PKG + ".solver_fdtd_hip")
P = importlib.import_module(PKG + ".params").PatchAntennaParams
tmp = tempfile.mkdtemp()
p245 = P.from_user_units(frequency_ghz=2.45, er=4.3, h_mm=1.6, loss_tangent=0.02)
p58 = P.from_user_units(frequency_ghz=5.8, er=4.3, h_mm=1.6, loss_tangent=0.02)
cases = [
    ("fixed 2.45", lambda: s.prepare_hip_patch_fixed(p245, work_dir=os.path.join(tmp, "a")), 2.45e9),
    ("microstrip 2.45", lambda: s.prepare_hip_microstrip_patch(p245, work_dir=os.path.join(tmp, "b")), 2.45e9),
    ("microstrip_3d 5.8 PML_8", lambda: s.prepare_hip_microstrip_patch_3d(p58, boundary="PML_8", work_dir=os.path.join(tmp, "c")), 5.8e9),
    ("microstrip_3d 2.45 MUR", lambda: s.prepare_hip_microstrip_patch_3d(p245, work_dir=os.path.join(tmp, "d")), 2.45e9),
    ("multi_3d 2x2 2.45", lambda: s.prepare_hip_microstrip_multi_3d(
        [s.PatchInstance(f"P{n}", p245, (ix - 0.5) * 0.0612, (iy - 0.5) * 0.0612, 0.0, s.FeedDirection.NEG_X)
         for n, (ix, iy) in enumerate([(0, 0), (1, 0), (0, 1), (1, 1)])], work_dir=os.path.join(tmp, "e")), 2.45e9),
    ("legacy 2.45", lambda: s.prepare_hip_patch(p245, work_dir=os.path.join(tmp, "f")), 2.45e9),
]
for name, prep_fn, f in cases:
    t0 = time.perf_counter()
    with warnings.catch_warnings(record=True) as ws:
        warnings.simplefilter("always")
        prep = prep_fn()
        if not prep.ok:
            print(name, "PREPARE FAILED", prep.message, flush=True); continue
        res = s.run_prepared_hip(prep, frequency_hz=f, verbose=0)
    dt = time.perf_counter() - t0
    if not res.ok:
        print(name, "RUN FAILED", res.message, flush=True); continue
    st = res.stats
    s11 = getattr(res, "s11_dB", None)
    line = f"{name:26s} grid {st['grid']} steps {st['steps']:6d} energy {st['energy_db']:7.1f} dB  {st['mcells_per_s']/1e3:6.1f} Gcells/s  call {dt:5.2f} s  Dmax {10*np.log10(res.Dmax):6.2f} dBi  max intensity {np.max(res.intensity):6.2f}"
    if s11 is not None:
        k = int(np.argmin(s11)); line += f"  S11 min {s11[k]:6.1f} dB at {res.freq[k]/1e9:.3f} GHz (f_res {res.f_res/1e9:.3f})"
        try:      # power balance at the S11 minimum: radiated (flux through the NF2FF box) over accepted at the first port
            fr = float(res.freq[k])
            nfr = prep.nf.CalcNF2FF(prep.sim_path, [fr], np.arange(0.0, 181.0, 6.0), np.arange(0.0, 360.0, 12.0), center=[0, 0, 0])
            prep.port.CalcPort(prep.sim_path, np.array([fr]))
            acc = sum(float(p.CalcPort(prep.sim_path, np.array([fr])).P_acc[0]) for p in (prep.ports or [prep.port]))
            line += f"  Prad/P_acc {100 * float(np.asarray(nfr.Prad)[0]) / acc:5.1f} %"
        except Exception as exc:      # noqa: BLE001
            line += f"  (power balance: {exc})"
    line += f"  warnings: {[str(w.message)[:60] for w in ws]} {st.get('nf2ff_warning')}"
    print(line, flush=True)

"""Round 4, GPU: launches that cannot abort the process (occupancy caps sized from the kernels' own LDS), and other robustness checks."""
import numpy as np
import pytest

from conftest import pkg
from helpers import patch_sim, seeded_fields

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("boundary", ["PEC", "CPML"])
@pytest.mark.parametrize("knob", ["FDTD_OCC_WF", "FDTD_OCC_E", "FDTD_OCC_H"])
@pytest.mark.parametrize("cap", ["1", "3"])
def test_occupancy_caps_run_or_return_an_error_code(hip_lib, monkeypatch, boundary, knob, cap):
    """An occupancy cap pads the dynamic LDS of a launch; round 3 sized the padding from hand-kept estimates of the kernels' static LDS and a cap
    of 1 on a kernel without CPML staging asked for more than a workgroup may have — the runtime ABORTED the process.  The padding now comes
    from hipFuncGetAttributes and the device's limits: every cap either runs (same fields as without it) or comes back as an error code."""
    capi = pkg("_capi")
    flags = capi.FLAG_KERNEL_DIRECT if knob != "FDTD_OCC_WF" else capi.FLAG_KERNEL_WAVEFRONT

    def run(env):
        if env:
            monkeypatch.setenv(knob, cap)
        s = patch_sim(60, 52, 34, boundary=boundary, cpml_cells=6, nr_ts=60, nf2ff=False)
        e = s.build(hip_lib, flags=flags)
        if env:
            monkeypatch.delenv(knob)
        seeded_fields(e, 3)
        e.run(60)
        return e.fields()
    ref = run(False)
    try:
        got = run(True)
    except capi.FdtdError as exc:          # an error code with a message is a legitimate outcome; an abort is not
        assert "LDS" in str(exc) or "launch" in str(exc)
        return
    assert np.array_equal(ref, got)


def _wide_port_sim(boundary, nr_ts=200):
    """The fixed patch scene with a lumped port as WIDE as the reference's multi-patch scene draws them (a box of many cells in x and y:
    solver_fdtd_openems_microstrip_multi_3d.py:472-541 — 1 350 source edges per port there): hundreds of source edges per strip-plane."""
    wl, sc, sim = pkg("workloads"), pkg("scene"), pkg("simulation")
    w = wl.patch_workload("test", nx=72, ny=66, nz=34)
    ports = [p for p in w.scene.ports]
    assert len(ports) == 1
    u = 1e-3
    w.scene.ports.clear()
    w.scene.add_lumped_port(1, 50.0, [-20, -12, 0], [14, 12, 1.6], "z", 1.0, priority=5)
    vox = sc.voxelize(w.scene, w.grid)
    nsrc = sum(p.src_idx.size for p in vox.ports)
    assert nsrc > 300, nsrc
    return sim.Simulation(w.grid, vox, f0=w.f0, fc=w.fc, boundary=boundary, cpml_cells=6, nr_ts=nr_ts, nf2ff_freqs=None)


@pytest.mark.parametrize("boundary,flag", [("CPML", "AUTO"), ("CPML", "DIRECT"), ("PEC", "WAVEFRONT"), ("MUR", "DIRECT"), ("MUR", "AUTO")])
def test_many_source_edges_equal_the_oracle(hip_lib, oracle_lib, boundary, flag):
    """Strip-planes with more source edges than a thread can scan take the dense LDS image (body_E); MUR AUTO: the resident schedule
    refuses a tile with more than 128 source edges and AUTO falls back to three launches.  Fields bit for bit against the oracle."""
    capi = pkg("_capi")
    flags = {"AUTO": 0, "DIRECT": capi.FLAG_KERNEL_DIRECT, "WAVEFRONT": capi.FLAG_KERNEL_WAVEFRONT}[flag]
    res = []
    for lib, fl in ((hip_lib, flags), (oracle_lib, 0)):
        s = _wide_port_sim(boundary)
        e = s.build(lib, flags=fl)
        e.run(120)       # the pulse is on: every source edge adds
        e.run(80)
        res.append((s, e))
    (sh, eh), (so, eo) = res
    fh, fo = eh.fields(), eo.fields()
    assert np.abs(fo).max() > 0 and np.array_equal(fh, fo)
    (uh, ih), (uo, io) = sh.port_series()[0], so.port_series()[0]
    assert np.abs(uo).max() > 0 and np.allclose(uh, uo, rtol=1e-12, atol=0) and np.allclose(ih, io, rtol=1e-12, atol=1e-300)


@pytest.mark.parametrize("flag", ["AUTO", "DIRECT", "WAVEFRONT"])
def test_discrete_poynting_theorem_on_the_gpu(hip_lib, flag):
    """The exact energy balance of the scheme (tests/test_oracle_invariants_cpu.py), stepped one timestep at a time by libfdtd_hip.so under the
    resident, the two-launch and the flag-coupled schedule: every timestep's balance closes to float32 round-off."""
    import test_oracle_invariants_cpu as inv
    capi = pkg("_capi")
    sim = inv._scene()
    e = sim.build(hip_lib, flags={"AUTO": 0, "DIRECT": capi.FLAG_KERNEL_DIRECT, "WAVEFRONT": capi.FLAG_KERNEL_WAVEFRONT}[flag])
    worst, q0, q_end = inv._identity(hip_lib, e, sim, 60, False)
    assert e.schedule_info()["resident"] == (flag == "AUTO")
    assert q_end < (1 - 5e-6) * q0 and worst < 1e-7, worst


@pytest.mark.parametrize("flag", ["AUTO", "DIRECT"])
def test_numerical_dispersion_relation_on_the_gpu(hip_lib, flag):
    """A cavity eigenmode stepped by libfdtd_hip.so follows the Yee dispersion relation of the GRID (1-2 % below the continuum's frequency for this
    mode) to float32 round-off: tests/test_oracle_invariants_cpu.py::_dispersion."""
    import test_oracle_invariants_cpu as inv
    capi = pkg("_capi")
    inv._dispersion(hip_lib, False, flags={"AUTO": 0, "DIRECT": capi.FLAG_KERNEL_DIRECT}[flag])


@pytest.mark.parametrize("shape", [(48, 44, 30), (61, 37, 23), (132, 70, 26)])
@pytest.mark.parametrize("kinds", [["MUR"] * 6, ["MUR", "PEC", "CPML", "MUR", "MUR", "CPML"]])
def test_mur_without_an_apply_pass_equals_the_apply_pass_and_the_oracle(hip_lib, oracle_lib, monkeypatch, shape, kinds):
    """Mur faces beyond the resident schedule: update_H takes the boundary voltages from the candidates of the post pass (stored behind the
    voltage arrays) and runs the pre pass itself — two launches per timestep.  Same fields, bit for bit, as with the apply pass as a launch
    of its own (FDTD_MUR_APPLY_PASS=1: three launches) and as the oracle; port series and recorded NF2FF faces alike."""
    monkeypatch.setenv("FDTD_RESIDENT", "0")
    out = []
    for lib, env in ((hip_lib, None), (hip_lib, "1"), (oracle_lib, None)):
        if env:
            monkeypatch.setenv("FDTD_MUR_APPLY_PASS", env)
        s = patch_sim(*shape, boundary=kinds, cpml_cells=5, nr_ts=240, nf2ff_mode="record")
        e = s.build(lib, flags=pkg("_capi").FLAG_KERNEL_DIRECT if lib is hip_lib else 0)   # (AUTO would take ONE launch where there are CPML layers too)
        if env:
            monkeypatch.delenv("FDTD_MUR_APPLY_PASS")
        seeded_fields(e, 11)
        for n in (1, 2, 77, 160):
            e.run(n)
        out.append((s, e))
    (s2, e2), (s3, e3), (so, eo) = out
    assert e2.schedule_info()["launches_per_timestep"] == 2 and e3.schedule_info()["launches_per_timestep"] == 3
    f2, f3, fo = e2.fields(), e3.fields(), eo.fields()
    assert np.abs(fo).max() > 0 and np.array_equal(f2, fo) and np.array_equal(f3, fo)
    for sa in (s2, s3):
        for (ua, ia), (ub, ib) in zip(sa.port_series(), so.port_series()):
            assert np.allclose(ua, ub, rtol=0, atol=1e-12 * np.abs(ub).max()) and np.allclose(ia, ib, rtol=0, atol=1e-12 * np.abs(ib).max())
        for a, b in zip(sa.nf2ff_boxes(), so.nf2ff_boxes()):
            assert np.abs(np.asarray(a) - np.asarray(b)).max() <= 1e-9 * np.abs(b).max()


def test_mur_probe_on_a_face_keeps_the_apply_pass(hip_lib, oracle_lib, monkeypatch):
    """Without the apply pass the boundary voltages in memory are the E update's own while update_H runs; a voltage probe that holds a node of
    a Mur face would sample them — such a scene keeps three launches per timestep (api.hip: mur_direct_possible), and equals the oracle."""
    monkeypatch.setenv("FDTD_RESIDENT", "0")
    out = []
    for lib in (hip_lib, oracle_lib):
        s = patch_sim(48, 44, 30, boundary="MUR", nr_ts=150, nf2ff=False)
        e = s.build(lib)
        nx, ny = 48, 44
        idx = np.array([(7 * ny + 9) * nx + 0, (7 * ny + 9) * nx + 1], dtype=np.int64)      # nodes (0, 9, 7) and (1, 9, 7): the first on the lower x face
        pid = e.add_probe(0, idx, np.array([1, 1], dtype=np.int8), np.array([1.0, 1.0], dtype=np.float32))
        seeded_fields(e, 5)
        e.run(150)
        out.append((e, e.get_probe(pid)))
    (eh, ph), (eo, po) = out
    assert eh.schedule_info()["launches_per_timestep"] == 3
    assert np.array_equal(eh.fields(), eo.fields())
    assert np.abs(po).max() > 0 and np.abs(ph - po).max() <= 1e-12 * np.abs(po).max()


@pytest.mark.parametrize("shape", [(6, 5, 5), (7, 6, 5), (9, 5, 7), (8, 9, 6), (13, 7, 9)])
@pytest.mark.parametrize("faces", [[1, 1, 1, 1, 1, 1], [1, 0, 0, 1, 1, 0], [0, 1, 1, 0, 0, 1]])
def test_mur_without_an_apply_pass_on_the_smallest_grids(hip_lib, oracle_lib, monkeypatch, shape, faces):
    """The smallest grids the two-launch Mur schedule takes (6 x 5 x 5 nodes: a boundary node, its inner neighbour and the opposite face's
    share one four-cell group, one row, one plane), every node next to a face, seeded fields, through the C ABI: fields identical to
    the oracle — and to the three-launch sequence."""
    capi, const = pkg("_capi"), pkg("constants")
    from opbuild_cases import random_scene
    monkeypatch.setenv("FDTD_RESIDENT", "0")
    grid, eps, kap, pec, _ = random_scene(5, shape, False, 3, n_lumped=0, pec_frac=0.0)
    dt = grid.courant_dt()
    nx, ny, nz = grid.shape
    out = []
    for lib, env in ((hip_lib, None), (hip_lib, "1"), (oracle_lib, None)):
        if env:
            monkeypatch.setenv("FDTD_MUR_APPLY_PASS", env)
        e = capi.Engine(lib, nx, ny, nz, dt, max_steps=64)
        if env:
            monkeypatch.delenv("FDTD_MUR_APPLY_PASS")
        eco = pkg("ecoperator")
        emet, hmet = eco.pack_metric_tables(*eco.metric_lists(grid, dt), grid)
        e.build_operator(grid.d, eps, kap, pec, const.EPS0, eco.lumped_overrides(grid, eps, kap, pec, dt, []), emet, hmet)
        coeff = []
        for f in range(6):
            l = grid.lines[f // 2]
            d = (l[-1] - l[-2]) if f % 2 else (l[1] - l[0])
            coeff.append((const.C0 * dt - d) / (const.C0 * dt + d))
        e.set_mur(faces, coeff)
        e.set_signal(np.zeros(4))
        seeded_fields(e, 17)
        for m in (1, 2, 37):
            e.run(m)
        out.append(e)
    e2, e3, eo = out
    assert e2.schedule_info()["launches_per_timestep"] == 2 and e3.schedule_info()["launches_per_timestep"] == 3
    fo = eo.fields()
    assert np.isfinite(fo).all() and np.abs(fo).max() > 0
    assert np.array_equal(e2.fields(), fo) and np.array_equal(e3.fields(), fo)


@pytest.mark.parametrize("transport", ["linked", "half_steps"])
@pytest.mark.parametrize("kinds", [["MUR"] * 6, ["MUR", "PEC", "CPML", "MUR", "MUR", "CPML"]])
def test_slabs_with_mur_faces_equal_one_slab(hip_lib, monkeypatch, transport, kinds):
    """Mur faces on a decomposed grid (x / y faces on every slab, z faces on the end slabs): the pre / post / apply launches with their state
    behind the voltage arrays (round 4 layout) — three slabs coupled by event-ordered peer copies, or stepped half-step by half-step with the
    halos carried by the host, reproduce ONE slab (two launches per timestep, no apply pass) bit for bit."""
    capi = pkg("_capi")
    monkeypatch.setenv("FDTD_RESIDENT", "0")
    shape, n, world = (50, 46, 33), 180, 3
    s1 = patch_sim(*shape, boundary=kinds, cpml_cells=5, nr_ts=n)
    e1 = s1.build(hip_lib)
    sims = [patch_sim(*shape, boundary=kinds, cpml_cells=5, nr_ts=n) for _ in range(world)]
    engs = [s.build(hip_lib, rank=r, world=world) for r, s in enumerate(sims)]
    rng = np.random.default_rng(9)
    for kind in (0, 1):
        for comp in range(3):
            g = (1e-3 * rng.standard_normal(e1.local_shape)).astype(np.float32)
            e1.set_field(kind, comp, g)
            for e in engs:
                e.set_field(kind, comp, np.ascontiguousarray(g[e.k0:e.k0 + e.nk]))
    e1.run(n)
    if transport == "linked":
        for m in (1, 60, n - 61):
            capi.run_linked(engs, m)
    else:
        for r in range(world - 1):       # the H halo of "step -1": the initial fields
            engs[r + 1].halo_put(capi.HALO_H_UP, engs[r].halo_get(capi.HALO_H_UP))
        for _ in range(n):
            for e in engs:
                e.half_step(capi.PHASE_E)
            for r in range(world - 1):
                engs[r].halo_put(capi.HALO_E_DOWN, engs[r + 1].halo_get(capi.HALO_E_DOWN))
            for e in engs:
                e.half_step(capi.PHASE_H)
            for r in range(world - 1):
                engs[r + 1].halo_put(capi.HALO_H_UP, engs[r].halo_get(capi.HALO_H_UP))
    f1, f2 = e1.fields(), np.concatenate([e.fields() for e in engs], axis=2)
    assert np.abs(f1).max() > 0 and np.array_equal(f1, f2)
    u1 = s1.port_series()[0][0]
    u2 = sum(s.port_series()[0][0] for s in sims)
    assert np.abs(u1).max() > 0 and np.abs(u2 - u1).max() <= 1e-12 * np.abs(u1).max()


@pytest.mark.parametrize("shape", [(48, 44, 30), (61, 37, 23), (132, 70, 26), (150, 140, 36)])
@pytest.mark.parametrize("kinds", [["MUR"] * 6, ["MUR", "PEC", "CPML", "MUR", "MUR", "CPML"]])
@pytest.mark.parametrize("multi", ["64", "1", "3"])
def test_mur_inside_the_one_launch_schedule_equals_the_oracle(hip_lib, oracle_lib, monkeypatch, shape, kinds, multi):
    """Mur faces inside k_step (one launch per timestep / several timesteps per launch): post pass in the E blocks, candidates read by the H blocks
    behind the flags of every E block that wrote one (wf_wait_mur), two candidate copies alternating with the timestep, pre pass and write-back by
    the H blocks with write-through stores.  Fields bit for bit, port series and recorded NF2FF faces against the oracle; cut launches."""
    capi = pkg("_capi")
    monkeypatch.setenv("FDTD_RESIDENT", "0")
    monkeypatch.setenv("FDTD_WF_MULTI", multi)
    out = []
    for lib in (hip_lib, oracle_lib):
        s = patch_sim(*shape, boundary=kinds, cpml_cells=5, nr_ts=260, nf2ff_mode="record")
        e = s.build(lib, flags=capi.FLAG_KERNEL_WAVEFRONT if lib is hip_lib else 0)
        seeded_fields(e, 21)
        for n in (1, 2, 90, 150):
            e.run(n)
        out.append((s, e))
    (sh, eh), (so, eo) = out
    info = eh.schedule_info()
    assert info["launches_per_timestep"] == 1 and not info["resident"]
    fh, fo = eh.fields(), eo.fields()
    assert np.abs(fo).max() > 0 and np.array_equal(fh, fo)
    for (ua, ia), (ub, ib) in zip(sh.port_series(), so.port_series()):
        assert np.allclose(ua, ub, rtol=0, atol=1e-12 * np.abs(ub).max()) and np.allclose(ia, ib, rtol=0, atol=1e-12 * np.abs(ib).max())
    for a, b in zip(sh.nf2ff_boxes(), so.nf2ff_boxes()):
        assert np.abs(np.asarray(a) - np.asarray(b)).max() <= 1e-9 * np.abs(b).max()


def test_microstrip_3d_mur_scene_one_launch_gpu_vs_oracle(hip_lib, oracle_lib, tmp_path):
    """The reference's 3-D microstrip patch at 5.8 GHz with its default MUR faces (167x143x101: beyond the resident schedule and above 1700 blocks
    per sweep, i.e. Mur faces INSIDE the one-launch schedule) through the plugin surface, 900 timesteps, on the HIP library and on the oracle:
    port series identical, S11 and pattern cuts within 1e-3 (tests/acceptance_mur_scenes.py runs it to its end: profiles/r04)."""
    s = pkg("solver_fdtd_hip")
    P = pkg("params").PatchAntennaParams
    p = P.from_user_units(frequency_ghz=5.8, er=4.3, h_mm=1.6, loss_tangent=0.02)
    out = []
    for lib, tag in ((hip_lib, "gpu"), (oracle_lib, "cpu")):
        prep = s.prepare_hip_microstrip_patch_3d(p, work_dir=str(tmp_path / tag), lib=lib)
        assert prep.ok, prep.message
        prep.FDTD.NrTS = 900
        r = s.run_prepared_hip(prep, frequency_hz=5.8e9, verbose=0)
        assert r.ok, r.message
        info = prep.FDTD.sim.engine.schedule_info() if lib is hip_lib else None
        out.append((r, prep.FDTD.sim.port_series()[0], s.s11_from_port(prep.port, prep.sim_path, 5.8e9)[1], info))
    (g, sg, s11g, info), (c, sc_, s11c, _) = out
    assert info["launches_per_timestep"] == 1 and not info["resident"]
    assert g.stats["grid"] == c.stats["grid"] and g.stats["steps"] == c.stats["steps"] == 900
    assert np.abs(sc_[0]).max() > 0 and np.array_equal(sg[0], sc_[0]) and np.array_equal(sg[1], sc_[1])
    assert np.linalg.norm(s11g - s11c) <= 1e-3 * np.linalg.norm(s11c)
    assert np.linalg.norm(g.intensity - c.intensity) <= 1e-3 * np.linalg.norm(c.intensity)


def test_mur_one_launch_flag_timeout_heals_itself(hip_lib, monkeypatch):
    """The bounded flag waits of the one-launch schedule with Mur faces (wf_wait_mur / wf_wait): under the fault hook (the H blocks of timestep 37 wait
    for a flag value nobody publishes, limit 20 us) the C ABI reports FDTD_E_DEVICE, and Simulation.run repeats the run under two launches per
    timestep — same fields and port series as a run that took two launches from the start."""
    capi = pkg("_capi")
    monkeypatch.setenv("FDTD_RESIDENT", "0")
    ref = patch_sim(132, 70, 26, boundary="MUR", nr_ts=300)
    e = ref.build(hip_lib, flags=capi.FLAG_KERNEL_DIRECT)
    ref.run(check_every=100)
    f_ref = e.fields()
    monkeypatch.setenv("FDTD_WF_FAULT_STEP", "37")
    s = patch_sim(132, 70, 26, boundary="MUR", nr_ts=300)
    eng = s.build(hip_lib, flags=capi.FLAG_KERNEL_WAVEFRONT)
    assert eng.schedule_info()["launches_per_timestep"] == 1
    with pytest.raises(capi.FdtdError, match="wavefront schedule"):
        eng.run(100)
    s = patch_sim(132, 70, 26, boundary="MUR", nr_ts=300)
    s.build(hip_lib, flags=capi.FLAG_KERNEL_WAVEFRONT)
    logs = []
    st = s.run(check_every=100, log=logs.append)
    assert st.schedule_fallback and "wavefront schedule" in st.schedule_fallback and logs
    assert st.steps == 300 and s.engine.schedule_info()["launches_per_timestep"] == 2
    assert np.array_equal(s.engine.fields(), f_ref)
    assert np.array_equal(s.port_series()[0][0], ref.port_series()[0][0])

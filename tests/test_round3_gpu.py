"""GPU tests added in round 3: the legacy plugin variant run on the GPU against the oracle, the self-healing step
schedule, the deterministic energy sum, Mur on a 5-node-wide grid, loop-back timing slabs at the ends of a decomposition,
uneven (cost-weighted) and mixed-schedule slabs, link / schedule introspection, the far-field frequency rules."""
import numpy as np
import pytest

from conftest import pkg
from helpers import patch_sim, seeded_fields, rel_l2

pytestmark = pytest.mark.gpu


def _params():
    P = pkg("params").PatchAntennaParams
    return P.from_user_units(frequency_ghz=2.45, er=4.3, h_mm=1.6, loss_tangent=0.02)


def test_legacy_variant_gpu_vs_oracle(hip_lib, oracle_lib, tmp_path):
    """prepare_hip_patch + run_prepared_hip(variant "legacy") — substrate and ground filling
    the box into the CPML layers ([3]*6), radian theta / phi (91 x 181), the attribute-sniffing conversion
    (solver_fdtd_openems.py:271-411) — on the GPU and on the oracle: same grid, same S11, patterns within 1e-3."""
    s = pkg("solver_fdtd_hip")
    res = []
    for lib, tag in ((hip_lib, "gpu"), (oracle_lib, "cpu")):
        prep = s.prepare_hip_patch(_params(), work_dir=str(tmp_path / tag), lib=lib)
        assert prep.ok, prep.message
        assert prep.variant == "legacy" and prep.FDTD.NrTS == 60000 and prep.FDTD.EndCriteria == 1e-5
        prep.FDTD.NrTS = 1500
        r = s.run_prepared_hip(prep, frequency_hz=2.45e9, verbose=0)
        assert r.ok, r.message
        res.append((r, prep))
    (g, pg), (c, pc) = res
    assert g.stats["grid"] == c.stats["grid"] and g.stats["steps"] == c.stats["steps"] == 1500
    assert pg.FDTD.sim.bc.kinds == ("CPML",) * 6
    assert g.is_dBi and g.intensity.shape == (91, 181) and g.theta[-1] == np.pi and abs(g.phi[-1] - 2 * np.pi) < 1e-15
    assert abs(g.intensity.max() - 10 * np.log10(g.Dmax)) < 1e-9       # the P_rad / Prad branch: max directivity = Dmax
    lin = lambda d: 10.0 ** (np.asarray(d) / 10.0)
    assert rel_l2(lin(g.intensity), lin(c.intensity)) < 1e-3
    assert rel_l2(g.s11, c.s11) < 1e-3 and rel_l2(g.port_u, c.port_u) < 1e-6
    assert abs(g.Dmax - c.Dmax) < 1e-3 * c.Dmax
    assert g.f_pattern == 2.45e9                                          # evaluated where the caller asked


def test_far_field_frequency_rules(hip_lib, tmp_path):
    """Default: every variant evaluates the far field at frequency_hz (what the reference effectively does — its
    resonance search, microstrip.py:407-433, is unreachable).  Opt-in resonance rule with running-DFT faces at
    caller-named frequencies: falls back to frequency_hz instead of failing."""
    s = pkg("solver_fdtd_hip")
    p = _params()
    prep = s.prepare_hip_microstrip_patch(p, feed_direction=s.FeedDirection.NEG_X, boundary="MUR", work_dir=str(tmp_path / "a"))
    assert prep.ok and prep.pattern_at_resonance is None
    prep.FDTD.NrTS = 9000
    r = s.run_prepared_hip(prep, frequency_hz=p.frequency_hz, verbose=0)
    assert r.ok, r.message
    assert r.s11_dB.min() < -10.0 and r.f_res != p.frequency_hz        # a dip exists, and still:
    assert r.f_pattern == p.frequency_hz
    prep = s.prepare_hip_microstrip_patch(p, feed_direction=s.FeedDirection.NEG_X, boundary="MUR", work_dir=str(tmp_path / "b"),
                                          nf2ff_mode="dft", nf2ff_freqs=[p.frequency_hz])
    prep.FDTD.NrTS = 9000
    prep.pattern_at_resonance = True
    r2 = s.run_prepared_hip(prep, frequency_hz=p.frequency_hz, verbose=0)
    assert r2.ok, r2.message
    assert r2.f_res == r.f_res and r2.f_pattern == p.frequency_hz
    assert rel_l2(10.0 ** (r2.intensity / 20), 10.0 ** (r.intensity / 20)) < 1e-3     # recorder == running DFT at the same frequency


def test_schedule_timeout_heals_itself(hip_lib, monkeypatch):
    """A flag wait of the one-launch schedule that times out (test hook: at step 37 the H blocks wait for a flag value
    nobody publishes, limit 20 us) -> the C ABI reports FDTD_E_DEVICE and clears the error word; Simulation.run rebuilds
    the context under the two-launch schedule and repeats the run from the initial state: same result as a DIRECT run."""
    capi = pkg("_capi")
    ref = patch_sim(64, 60, 36, nr_ts=300)
    e = ref.build(hip_lib, flags=capi.FLAG_KERNEL_DIRECT)
    st_ref = ref.run(check_every=100)
    f_ref = e.fields()
    monkeypatch.setenv("FDTD_WF_FAULT_STEP", "37")
    monkeypatch.setenv("FDTD_RESIDENT", "0")      # (round 4: a grid this small steps resident in registers under AUTO — its own fault test: test_resident_gpu.py)
    s = patch_sim(64, 60, 36, nr_ts=300)
    eng = s.build(hip_lib)
    assert eng.schedule_info()["launches_per_timestep"] == 1 and not eng.schedule_info()["resident"]
    with pytest.raises(capi.FdtdError, match="wavefront schedule"):
        eng.run(100)
    eng.run(10)                                   # the error word was cleared: the context is usable (its fields are not)
    s = patch_sim(64, 60, 36, nr_ts=300)
    s.build(hip_lib)
    logs = []
    st = s.run(check_every=100, log=logs.append)
    assert st.schedule_fallback and "wavefront schedule" in st.schedule_fallback and logs
    assert st.steps == 300 and s.engine.schedule_info()["launches_per_timestep"] == 2
    assert np.array_equal(s.engine.fields(), f_ref) and st.energy_db == st_ref.energy_db
    u, i = s.port_series()[0]
    u0, i0 = ref.port_series()[0]
    assert np.array_equal(u, u0) and np.array_equal(i, i0)


def test_energy_sum_is_reproducible(hip_lib, oracle_lib):
    """fdtd_energy: per-block partials added in block order by the last block (no atomicAdd on doubles): identical bits
    from call to call and from context to context; equal to the oracle's sums to 1e-12."""
    vals = []
    for _ in range(2):
        s = patch_sim(53, 47, 31, nr_ts=120)
        e = s.build(hip_lib)
        seeded_fields(e, 3)
        e.run(120)
        vals.append([e.energy() for _ in range(3)])
    flat = [v for run in vals for v in run]
    assert all(v == flat[0] for v in flat)
    s = patch_sim(53, 47, 31, nr_ts=120)
    e = s.build(oracle_lib)
    seeded_fields(e, 3)
    e.run(120)
    assert np.allclose(flat[0], e.energy(), rtol=1e-12)


@pytest.mark.parametrize("shape", [(5, 20, 18), (6, 20, 18), (20, 5, 18), (20, 18, 5)])
def test_mur_on_five_node_wide_grids(hip_lib, oracle_lib, shape):
    """Mur faces on a grid 5 nodes wide: along x both inner nodes (1 and 3) lie in ONE thread's four cells, so the post
    pass fused into update_E (one x pair of S values per thread) must not be taken there; 6 wide it is."""
    capi, const = pkg("_capi"), pkg("constants")
    from opbuild_cases import random_scene
    grid, eps, kap, pec, _ = random_scene(5, shape, False, 3, n_lumped=0, pec_frac=0.0)
    dt = grid.courant_dt()
    nx, ny, nz = grid.shape
    n = 90
    out = []
    for lib in (hip_lib, oracle_lib):
        e = capi.Engine(lib, nx, ny, nz, dt, max_steps=n + 8)
        eco = pkg("ecoperator")
        emet, hmet = eco.pack_metric_tables(*eco.metric_lists(grid, dt), grid)
        e.build_operator(grid.d, eps, kap, pec, const.EPS0, eco.lumped_overrides(grid, eps, kap, pec, dt, []), emet, hmet)
        coeff = []
        for f in range(6):
            l = grid.lines[f // 2]
            d = (l[-1] - l[-2]) if f % 2 else (l[1] - l[0])
            coeff.append((const.C0 * dt - d) / (const.C0 * dt + d))
        e.set_mur([1] * 6, coeff)
        t = np.arange(n) * dt
        e.set_signal(np.sin(2 * np.pi * 8e9 * t) * np.exp(-((t - 30 * dt) / (12 * dt)) ** 2))
        i, j, k = nx // 2, ny // 2, nz // 2
        comp = int(np.argmin(shape))                       # drive the component along the thin axis
        e.add_source(np.array([(k * ny + j) * nx + i], np.int64), np.array([comp], np.int8), np.array([1.0], np.float32))
        seeded_fields(e, 11, 1e-4)
        e.run(n)
        out.append(e.fields())
    assert np.abs(out[1]).max() > 0 and np.array_equal(out[0], out[1]), f"rel L2 {rel_l2(out[0], out[1]):.3e}"


def test_loopback_slabs_at_both_ends_of_a_decomposition(hip_lib):
    """tools/slab_balance.py times every slab of a decomposition alone on one GPU: FDTD_FLAG_LOOPBACK + its own blob for
    both neighbours.  Rank 0, an interior rank and the last rank all step (no halo wait times out), under both schedules;
    without the flag an end rank refuses a blob for the neighbour it does not have."""
    capi = pkg("_capi")
    world = 4
    for flags in (capi.FLAG_KERNEL_DIRECT, capi.FLAG_KERNEL_WAVEFRONT):
        for rank in (0, 2, world - 1):
            s = patch_sim(56, 52, 48, nr_ts=80, nf2ff=False)
            e = s.build(hip_lib, rank=rank, world=world, flags=flags | capi.FLAG_LOOPBACK)
            blob = e.p2p_export()
            e.p2p_attach(blob, blob)
            seeded_fields(e, rank)
            e.run(60)
            assert e.step == 60 and e.schedule_info()["transport"] == "p2p"
            li = e.p2p_link_info(1)
            assert li["same_device"] and li["mapping"] == "same-process" and li["device"] == 0
    s = patch_sim(56, 52, 48, nr_ts=80, nf2ff=False)
    e = s.build(hip_lib, rank=0, world=world)
    blob = e.p2p_export()
    with pytest.raises(capi.FdtdError, match="exactly its existing neighbours"):
        e.p2p_attach(blob, blob)


def _attach_p2p(engs):
    blobs = [e.p2p_export() for e in engs]
    for r, e in enumerate(engs):
        e.p2p_attach(blobs[r - 1] if r > 0 else None, blobs[r + 1] if r + 1 < len(engs) else None)


@pytest.mark.parametrize("world", [3, 4])
def test_cost_weighted_slabs_and_mixed_schedules_equal_one_slab(hip_lib, world):
    """The cost-weighted z-partition (end ranks own fewer planes: all z-CPML planes are theirs) on the mailbox transport,
    the slabs stepping under DIFFERENT schedules (one launch per timestep on the even ranks, two on the odd ones — what
    AUTO does when only some slabs of an uneven partition pass its block-count threshold): bit-identical to one slab."""
    capi, simm = pkg("_capi"), pkg("simulation")
    s1 = patch_sim(56, 52, 60, cpml_cells=10, nr_ts=220)
    e1 = s1.build(hip_lib, flags=capi.FLAG_KERNEL_DIRECT)
    sims = [patch_sim(56, 52, 60, cpml_cells=10, nr_ts=220) for _ in range(world)]
    flags = [capi.FLAG_KERNEL_WAVEFRONT if r % 2 == 0 else capi.FLAG_KERNEL_DIRECT for r in range(world)]
    engs = [s.build(hip_lib, rank=r, world=world, flags=flags[r]) for r, s in enumerate(sims)]
    nks = [e.nk for e in engs]
    assert sum(nks) == 60 and nks[0] < max(nks) and nks[-1] < max(nks)             # uneven: the end slabs are thinner
    assert nks != [simm.slab_range(60, world, r)[1] for r in range(world)]
    assert [e.schedule_info()["launches_per_timestep"] for e in engs] == [0] * world   # not steppable before a transport is attached
    _attach_p2p(engs)
    assert [e.schedule_info()["launches_per_timestep"] for e in engs] == [1 if r % 2 == 0 else 2 for r in range(world)]
    rng = np.random.default_rng(9)
    for kind in (0, 1):
        for comp in range(3):
            g = (1e-3 * rng.standard_normal(e1.local_shape)).astype(np.float32)
            e1.set_field(kind, comp, g)
            for e in engs:
                e.set_field(kind, comp, np.ascontiguousarray(g[e.k0:e.k0 + e.nk]))
    e1.run(220)
    for n in (1, 90, 129):
        capi.run_linked(engs, n)
    f1, f2 = e1.fields(), np.concatenate([e.fields() for e in engs], axis=2)
    assert np.abs(f1).max() > 0 and np.array_equal(f1, f2)
    u1, i1 = s1.port_series()[0]
    assert rel_l2(sum(s.port_series()[0][0] for s in sims), u1) < 1e-12
    assert rel_l2(sum(s.port_series()[0][1] for s in sims), i1) < 1e-12


def test_schedule_info_tells_the_schedule(hip_lib, monkeypatch):
    capi = pkg("_capi")
    s = patch_sim(64, 60, 36, nr_ts=10)
    info = s.build(hip_lib).schedule_info()       # round 4: small grids (at most two tiles per CU) step resident in registers, whatever their faces
    assert info["resident"] and info["launches_per_timestep"] == 1 and info["lag_planes"] == -1 and info["blocks_per_sweep"] <= 512
    monkeypatch.setenv("FDTD_RESIDENT", "0")      # ... the schedules below are what AUTO takes where the resident one does not fit
    info = s.build(hip_lib).schedule_info()
    assert info["launches_per_timestep"] == 1 and info["lag_planes"] == 36 and info["transport"] == "none" and info["xcd_shares_weighted"]
    assert info["blocks_per_sweep"] > 0 and info["rows_per_strip"] >= 4
    s = patch_sim(64, 60, 36, nr_ts=10)
    assert s.build(hip_lib, flags=capi.FLAG_KERNEL_DIRECT).schedule_info()["launches_per_timestep"] == 2
    s = patch_sim(48, 44, 30, boundary="MUR", nr_ts=10)
    info = s.build(hip_lib).schedule_info()
    assert info["launches_per_timestep"] == 2 and not info["xcd_shares_weighted"] and not info["resident"]     # (round 4: no apply pass)
    monkeypatch.delenv("FDTD_RESIDENT")
    info = s.build(hip_lib).schedule_info()       # small Mur scene: resident in registers (round 4), one launch holds up to 256 timesteps
    assert info["resident"] and info["launches_per_timestep"] == 1 and info["timesteps_per_launch_max"] == 256
    monkeypatch.setenv("FDTD_RESIDENT", "0")
    s = patch_sim(48, 44, 30, boundary="PEC", nr_ts=10)
    assert s.build(hip_lib).schedule_info()["launches_per_timestep"] == 2     # small grid without CPML: two launches


def test_measured_xcd_shares_leave_the_results_alone(hip_lib, oracle_lib, monkeypatch):
    """The last launch of an fdtd_run call of >= 16 timesteps is a calibration launch (every block leaves its end time) and
    the shares of the eight XCDs are re-cut from the measured finish times: the cuts move between the calls of one run,
    the fields may not.  Ten calls on the GPU against one run on the oracle, and against a context that never adapts."""
    capi = pkg("_capi")
    monkeypatch.setenv("FDTD_RESIDENT", "0")      # (the XCD shares belong to the flag-coupled launches; round 4 steps a grid this small resident under AUTO)
    s = patch_sim(64, 60, 36, nr_ts=400)
    e = s.build(hip_lib)
    for _ in range(10):
        e.run(32)
    info = e.schedule_info()
    assert info["launches_per_timestep"] == 1 and info["xcd_adaptations"] >= 8
    so = patch_sim(64, 60, 36, nr_ts=400)
    eo = so.build(oracle_lib)
    eo.run(320)
    assert np.array_equal(e.fields(), eo.fields())
    assert rel_l2(s.port_series()[0][0], so.port_series()[0][0]) < 1e-12
    monkeypatch.setenv("FDTD_XCD_ADAPT", "0")
    s0 = patch_sim(64, 60, 36, nr_ts=400)
    e0 = s0.build(hip_lib)
    for _ in range(10):
        e0.run(32)
    assert e0.schedule_info()["xcd_adaptations"] == 0 and np.array_equal(e0.fields(), e.fields())


@pytest.mark.parametrize("shape,bc", [((64, 60, 36), "CPML"), ((53, 47, 31), "CPML"), ((260, 18, 40), "CPML"), ((64, 60, 36), "PEC")])
def test_several_timesteps_per_launch_equal_one_per_launch(hip_lib, oracle_lib, shape, bc, monkeypatch):
    """Cache-resident single slabs step SEVERAL timesteps per launch (k_step<.., MULTI>: the E blocks of timestep s + 1 wait
    for the flags of the H blocks of timestep s; device-scope loads and write-through stores for fields and psi; launches cut
    at the timesteps whose NF2FF faces are sampled).  Against one launch per timestep ($FDTD_WF_MULTI=1) and against the
    oracle: fields, port series (the probe blocks of every timestep inside the launch) and the recorded NF2FF faces, over
    calls of 1, 7, 100 and 61 timesteps from seeded fields."""
    capi = pkg("_capi")
    out = []
    for multi, lib in (("64", hip_lib), ("1", hip_lib), ("64", oracle_lib)):
        monkeypatch.setenv("FDTD_WF_MULTI", multi)
        s = patch_sim(*shape, boundary=bc, cpml_cells=6, nr_ts=200, nf2ff_mode="record")
        e = s.build(lib, flags=capi.FLAG_KERNEL_WAVEFRONT if lib is hip_lib else 0)
        if lib is hip_lib:
            assert e.schedule_info()["timesteps_per_launch_max"] == int(multi)
        seeded_fields(e, 17)
        for n in (1, 7, 100, 61):
            e.run(n)
        out.append((e.fields(), s.port_series()[0], s.nf2ff_boxes()))
    (fm, pm, bm), (f1, p1, b1), (fo, po, bo) = out
    assert np.abs(fo).max() > 0 and np.array_equal(fm, f1) and np.array_equal(fm, fo)
    assert np.array_equal(pm[0], p1[0]) and np.array_equal(pm[1], p1[1])
    assert rel_l2(pm[0], po[0]) < 1e-12 and rel_l2(pm[1], po[1]) < 1e-12
    for a, b, c in zip(bm, b1, bo):
        assert np.array_equal(a, b) and rel_l2(a, c) < 1e-12


# ---- GPU twins of the host-layer known-answer tests (tests/test_host_layer_kat_cpu.py): the same closed-form answers with
# ---- libfdtd_hip.so as the engine, and the engine's numbers equal to the oracle's
def test_microstrip_line_kat_on_the_gpu(hip_lib, oracle_lib, tmp_path):
    import test_host_layer_kat_cpu as kat
    pd = pkg("patch_design")
    f = np.linspace(1.0e9, 2.0e9, 11)
    res = {}
    for length in (30.0, 50.0):
        fdtd, p1, p2, W = kat._microstrip_line(hip_lib, str(tmp_path / f"g{int(length)}"), length, 4.3, 1.6, nr_ts=5000)
        res[length] = kat._abcd_of_line(p1, p2, f)
        if length == 30.0:      # engine parity on this scene (graded mesh, Mur faces, two lumped ports)
            _, o1, o2, _ = kat._microstrip_line(oracle_lib, str(tmp_path / "o30"), length, 4.3, 1.6, nr_ts=5000)
            zo, blo, _ = kat._abcd_of_line(o1, o2, f)
            assert rel_l2(p1.u_data.ui_val[0], o1.u_data.ui_val[0]) < 1e-9 and rel_l2(p2.i_data.ui_val[0], o2.i_data.ui_val[0]) < 1e-9
            assert rel_l2(res[30.0][0], zo) < 1e-6
    z0 = res[30.0][0]
    assert np.all(np.abs(z0.real - 50.0) < 2.5) and np.all(np.abs(z0.imag) < 1.0), z0
    eps_eff = ((res[50.0][1] - res[30.0][1]) / 20e-3 * kat.C0 / (2 * np.pi * f)) ** 2
    assert np.all(np.abs(eps_eff / pd.effective_eps(4.3, 1.6e-3, W * 1e-3) - 1.0) < 0.03), eps_eff


@pytest.mark.parametrize("half", [False, True])
def test_cpml_terminating_a_dielectric_on_the_gpu(hip_lib, half):
    import test_host_layer_kat_cpu as kat
    n, steps, off = 64, 700, 230
    small = kat._plate_line(hip_lib, n, n, "CPML", steps, (32, 32), (40, 27), 4.3, half)
    big = kat._plate_line(hip_lib, n + 2 * off, n + 2 * off, "PEC", steps, (32 + off, 32 + off), (40 + off, 27 + off), 4.3, half)
    err = np.max(np.abs(small - big)) / np.max(np.abs(big))
    assert 20 * np.log10(err) < -60.0, f"reflection {20 * np.log10(err):.1f} dB"


def test_randomised_cases_equal_the_oracle(hip_lib, oracle_lib):
    """A fixed-seed batch of tests/fuzz_parity.py: 48 drawn combinations of grid shape, boundary per face, layer thickness, operator
    form, schedule, tiling, timesteps per launch, NF2FF mode and run-call lengths — fields bit for bit, port series, energy and NF2FF
    face spectra against the oracle.  (1060 + 350 further cases of other seeds: profiles/r03/randomised_parity.txt.)"""
    import fuzz_parity
    lines = []
    failed = fuzz_parity.run_batch(48, 7, hip_lib, oracle_lib, log=lines.append)
    ran = [l for l in lines if ": ok " in l or ": FAIL" in l]
    assert len(ran) >= 40, "\n".join(lines)
    assert not failed, "\n".join(l for l in lines if "FAIL" in l)
    assert any("ts/launch 64" in l for l in ran) and any("launches/ts 2" in l for l in ran)
    # ... Mur faces on the launch-per-half-step schedules (round 4: two launches, the apply pass inside update_H; now and then three)
    lines = []
    failed = fuzz_parity.run_batch(24, 13, hip_lib, oracle_lib, log=lines.append, mur=True)
    ran = [l for l in lines if ": ok " in l or ": FAIL" in l]
    assert len(ran) >= 20 and not failed, "\n".join(lines)
    assert any("launches/ts 3" in l for l in ran) and any("launches/ts 2" in l for l in ran)
    # ... and decomposed runs: 2 ... 6 P2P slabs of drawn partition and per-slab schedule in this process against ONE slab on the oracle
    lines = []
    failed = fuzz_parity.run_batch(16, 11, hip_lib, oracle_lib, log=lines.append, slabs=True)
    assert len([l for l in lines if ": ok " in l]) >= 12 and not failed, "\n".join(lines)

"""GPU parity: libfdtd_hip.so vs the CPU oracle through the same C ABI, same inputs.

Bar (tightened): float32 fields IDENTICAL as IEEE values after hundreds of steps — the
HIP kernels and the oracle spell the same fmaf sequence, so every bit agrees except the sign of zero on
never-updated dead boundary edges (0 * out-of-range neighbour, which is the previous row in the dense
host layout and a pad cell in the device layout); probe/DFT/energy reductions (different summation
order) within 1e-12 relative (float64 accumulators)."""
import numpy as np
import pytest

from conftest import pkg
from helpers import patch_sim, seeded_fields, rel_l2


def same_values(a, b):
    """Exact float equality (-0 == +0; any NaN fails)."""
    return np.array_equal(a, b)

pytestmark = pytest.mark.gpu


def _run_both(sim_factory, hip_lib, oracle_lib, steps, seed=None, flags=0):
    out = []
    for lib in (hip_lib, oracle_lib):
        s = sim_factory()
        e = s.build(lib, flags=flags if lib is hip_lib else 0)
        if seed is not None:
            seeded_fields(e, seed)
        e.run(steps)
        out.append((s, e))
    return out


def _kflags(kernel):
    capi = pkg("_capi")
    return {"direct": capi.FLAG_KERNEL_DIRECT, "fused": capi.FLAG_KERNEL_FUSED, "tile": capi.FLAG_KERNEL_TILE,
            "march": capi.FLAG_KERNEL_MARCH}[kernel]


@pytest.mark.parametrize("kernel", ["fused", "tile", "march", "direct"])
@pytest.mark.parametrize("use_classes", [True, False])
@pytest.mark.parametrize("shape", [(64, 60, 36), (53, 47, 31)])
def test_fields_bitexact_cpml(hip_lib, oracle_lib, shape, use_classes, kernel):
    """kernel = fused: one-pass E+H sweep with ping-pong buffers (class operators only; the raw operator
    falls back to two passes under AUTO); direct: two-pass leapfrog."""
    capi = pkg("_capi")
    if kernel != "direct" and not use_classes:
        pytest.skip("the one-pass kernels need the class-compressed operator")
    flags = _kflags(kernel)
    (sh, eh), (so, eo) = _run_both(lambda: patch_sim(*shape, boundary="CPML", cpml_cells=8, nr_ts=300,
                                                     use_classes=use_classes), hip_lib, oracle_lib, 300, seed=1, flags=flags)
    assert eh.backend.startswith("hip") and eo.backend.startswith("oracle")
    fh, fo = eh.fields(), eo.fields()
    assert np.isfinite(fo).all() and np.abs(fo).max() > 0
    assert same_values(fh, fo), f"rel L2 {rel_l2(fh, fo):.3e}"
    nz = fo != 0
    assert np.array_equal(fh[nz].view(np.uint32), fo[nz].view(np.uint32))


def test_fields_bitexact_mur(hip_lib, oracle_lib):
    (sh, eh), (so, eo) = _run_both(lambda: patch_sim(48, 44, 30, boundary="MUR", nr_ts=400), hip_lib, oracle_lib, 400, seed=2)
    fh, fo = eh.fields(), eo.fields()
    assert same_values(fh, fo), f"rel L2 {rel_l2(fh, fo):.3e}"
    nz = fo != 0
    assert np.array_equal(fh[nz].view(np.uint32), fo[nz].view(np.uint32))


def test_fields_bitexact_mixed_boundaries(hip_lib, oracle_lib):
    bc = ["MUR", "CPML", "CPML", "MUR", "PEC", "CPML"]
    (sh, eh), (so, eo) = _run_both(lambda: patch_sim(50, 46, 33, boundary=bc, cpml_cells=6, nr_ts=300), hip_lib, oracle_lib, 300, seed=3)
    fh, fo = eh.fields(), eo.fields()
    assert same_values(fh, fo), f"rel L2 {rel_l2(fh, fo):.3e}"
    nz = fo != 0
    assert np.array_equal(fh[nz].view(np.uint32), fo[nz].view(np.uint32))


def test_port_probes_dft_energy(hip_lib, oracle_lib):
    (sh, eh), (so, eo) = _run_both(lambda: patch_sim(64, 60, 36, nr_ts=1500), hip_lib, oracle_lib, 1500)
    (uh, ih), (uo, io) = sh.port_series()[0], so.port_series()[0]
    assert len(uh) == len(uo) == 1500 and np.abs(uo).max() > 0
    assert rel_l2(uh, uo) < 1e-12 and rel_l2(ih, io) < 1e-12
    bh, bo = sh.nf2ff_boxes(), so.nf2ff_boxes()
    assert len(bh) == len(bo) == 24
    for a, b in zip(bh, bo):
        assert a.shape == b.shape
        assert rel_l2(a, b) < 1e-12
    evh, eih = eh.energy(); evo, eio = eo.energy()
    assert abs(evh - evo) <= 1e-10 * evo and abs(eih - eio) <= 1e-10 * eio


def test_fused_flag_rejects_ineligible_scene(hip_lib):
    capi = pkg("_capi")
    s = patch_sim(40, 40, 30, boundary="MUR", nr_ts=20, nf2ff=False)
    e = s.build(hip_lib, flags=capi.FLAG_KERNEL_FUSED)
    with pytest.raises(capi.FdtdError, match="fused kernel needs"):
        e.run(2)


@pytest.mark.parametrize("kernel", ["fused", "tile", "march"])
def test_fused_no_pml_and_odd_sizes(hip_lib, oracle_lib, kernel):
    """PEC box (no CPML template path), nx not a multiple of 4, sources + probes, one-pass kernels vs oracle."""
    capi = pkg("_capi")
    (sh, eh), (so, eo) = _run_both(lambda: patch_sim(45, 43, 29, boundary="PEC", nr_ts=400, nf2ff=False), hip_lib, oracle_lib, 400,
                                   flags=_kflags(kernel))
    assert same_values(eh.fields(), eo.fields()) and np.abs(eo.fields()).max() > 0
    assert rel_l2(sh.port_series()[0][0], so.port_series()[0][0]) < 1e-12


@pytest.mark.parametrize("kernel", ["fused", "tile", "march", "direct"])
def test_chunked_runs_and_fused_probes(hip_lib, oracle_lib, kernel):
    """fdtd_run in uneven chunks (probe flush at every call end, sources injected inside the main kernels)
    must give the same series as the oracle's plain loop."""
    capi = pkg("_capi")
    sh, so = patch_sim(48, 44, 32, nr_ts=700), patch_sim(48, 44, 32, nr_ts=700)
    eh = sh.build(hip_lib, flags=_kflags(kernel))
    eo = so.build(oracle_lib)
    for n in (1, 2, 97, 250, 349, 1):
        eh.run(n)
    eo.run(700)
    assert eh.step == eo.step == 700
    (uh, ih), (uo, io) = sh.port_series()[0], so.port_series()[0]
    assert len(uh) == 700 and np.abs(uo).max() > 0 and np.abs(io).max() > 0
    assert rel_l2(uh, uo) < 1e-12 and rel_l2(ih, io) < 1e-12
    assert same_values(eh.fields(), eo.fields())
    for a, b in zip(sh.nf2ff_boxes(), so.nf2ff_boxes()):
        assert rel_l2(a, b) < 1e-12


def test_farfield_matches_oracle(hip_lib, oracle_lib):
    capi = pkg("_capi")
    rng = np.random.default_rng(5)
    npts = 3000
    pos = rng.uniform(-0.05, 0.05, (npts, 3))
    Js = rng.standard_normal((npts, 3)) + 1j * rng.standard_normal((npts, 3))
    Ms = rng.standard_normal((npts, 3)) + 1j * rng.standard_normal((npts, 3))
    th, ph = np.meshgrid(np.deg2rad(np.arange(0, 181, 4.0)), np.deg2rad(np.arange(0, 360, 10.0)), indexing="ij")
    k = 2 * np.pi * 2.45e9 / 299792458.0
    eh = capi.farfield(hip_lib, pos, Js, Ms, k, th.ravel(), ph.ravel())
    eo = capi.farfield(oracle_lib, pos, Js, Ms, k, th.ravel(), ph.ravel())
    assert rel_l2(eh[0], eo[0]) < 1e-11 and rel_l2(eh[1], eo[1]) < 1e-11


def test_two_slabs_equal_one_slab(hip_lib):
    """z-slab decomposition with the external halo transport: two contexts on one device, halos
    copied by the host, must reproduce the single-slab run bit for bit (N-GPU == 1-GPU)."""
    capi = pkg("_capi")
    s1 = patch_sim(56, 52, 34, nr_ts=200)
    e1 = s1.build(hip_lib)
    e1.run(200)
    sa, sb = patch_sim(56, 52, 34, nr_ts=200), patch_sim(56, 52, 34, nr_ts=200)
    ea, eb = sa.build(hip_lib, rank=0, world=2), sb.build(hip_lib, rank=1, world=2)
    for _ in range(200):
        ea.half_step(capi.PHASE_E); eb.half_step(capi.PHASE_E)
        ea.halo_put(capi.HALO_E_DOWN, eb.halo_get(capi.HALO_E_DOWN))
        ea.half_step(capi.PHASE_H); eb.half_step(capi.PHASE_H)
        eb.halo_put(capi.HALO_H_UP, ea.halo_get(capi.HALO_H_UP))
    f1 = e1.fields()
    f2 = np.concatenate([ea.fields(), eb.fields()], axis=2)
    assert np.array_equal(f1.view(np.uint32), f2.view(np.uint32))  # same layout on both sides: every bit
    u1 = s1.port_series()[0][0]
    u2 = sa.port_series()[0][0] + sb.port_series()[0][0]
    assert rel_l2(u2, u1) < 1e-12


@pytest.mark.parametrize("overlap", ["split", "nosplit"])
@pytest.mark.parametrize("world", [2, 3, 5])
def test_linked_slabs_overlapped_schedule(hip_lib, world, overlap):
    """The multi-rank step schedule of the RCCL path (interior planes first, dependent boundary plane after
    the halo event, probes/sources fused into the split launches) driven in-process: `world` slabs on one
    GPU with peer copies instead of ncclSend/ncclRecv must reproduce the single-slab run bit for bit."""
    capi = pkg("_capi")
    s1 = patch_sim(56, 52, 34, nr_ts=260)
    e1 = s1.build(hip_lib)
    e1.run(260)
    sims = [patch_sim(56, 52, 34, nr_ts=260) for _ in range(world)]
    flag = capi.FLAG_OVERLAP_ON if overlap == "split" else capi.FLAG_OVERLAP_OFF
    engs = [s.build(hip_lib, rank=r, world=world, flags=flag) for r, s in enumerate(sims)]
    for n in (1, 100, 159):
        capi.run_linked(engs, n)
    f2 = np.concatenate([e.fields() for e in engs], axis=2)
    assert np.array_equal(e1.fields().view(np.uint32), f2.view(np.uint32))
    u1, i1 = s1.port_series()[0]
    u2 = sum(s.port_series()[0][0] for s in sims)
    i2 = sum(s.port_series()[0][1] for s in sims)
    assert rel_l2(u2, u1) < 1e-12 and rel_l2(i2, i1) < 1e-12
    for a, *parts in zip(s1.nf2ff_boxes(), *[s.nf2ff_boxes() for s in sims]):
        assert rel_l2(sum(parts), a) < 1e-12


def test_rccl_single_rank_comm(hip_lib):
    """World of one: the RCCL communicator initialises and the step loop runs through it."""
    capi = pkg("_capi")
    s = patch_sim(40, 40, 30, nr_ts=50, nf2ff=False)
    e = s.build(hip_lib)
    e.comm_init(capi.comm_unique_id(hip_lib))
    e.run(50)
    assert e.step == 50


def test_full_size_properties(hip_lib):
    """BASELINE 'NS' size (300x300x60): size-independent properties instead of an oracle run —
    zero in -> zero out, linearity in the excitation, finite decaying energy."""
    sim_m, wl, sc = pkg("simulation"), pkg("workloads"), pkg("scene")
    w = wl.patch_workload("NS")
    vox = sc.voxelize(w.scene, w.grid)
    res = []
    base_amp = [p.src_amp.copy() for p in vox.ports]
    for amp in (0.0, 1.0, 2.0):
        for p, a0 in zip(vox.ports, base_amp):
            p.src_amp = (amp * a0).astype(np.float32)
        s = sim_m.Simulation(w.grid, vox, f0=w.f0, fc=w.fc, boundary="CPML", cpml_cells=10, nr_ts=600)
        e = s.build(hip_lib)
        e.run(600)
        res.append((e.get_field(0, 2), s.port_series()[0][0]))
    for p, a0 in zip(vox.ports, base_amp):
        p.src_amp = a0
    assert not res[0][0].any() and not res[0][1].any()
    assert np.isfinite(res[1][0]).all() and np.abs(res[1][0]).max() > 0
    assert rel_l2(res[2][0], 2.0 * res[1][0]) < 1e-6
    assert rel_l2(res[2][1], 2.0 * res[1][1]) < 1e-6


def test_plugin_path_s11_and_patterns_gpu_vs_oracle(hip_lib, oracle_lib, tmp_path):
    """North-star parity statement, end to end through the plugin surface: the same PatchAntennaParams
    go through prepare_hip_microstrip_patch_3d / run_prepared_hip once on the GPU library and once on the
    CPU oracle; S11(f) and the E/H-plane patterns must agree to 1e-3 relative L2 (they agree to ~1e-9:
    the time stepping is bit-identical, only fp64 reductions differ)."""
    s = pkg("solver_fdtd_hip")
    sink = pkg("result_sink")
    P = pkg("params").PatchAntennaParams
    p = P.from_user_units(frequency_ghz=5.8, er=4.3, h_mm=1.6, loss_tangent=0.02)
    res = []
    for lib, tag in ((hip_lib, "gpu"), (oracle_lib, "cpu")):
        prep = s.prepare_hip_microstrip_patch_3d(p, feed_direction=s.FeedDirection.NEG_X, boundary="PML_8", theta_step_deg=4.0,
                                                 phi_step_deg=15.0, mesh_quality=1, work_dir=str(tmp_path / tag), lib=lib)
        assert prep.ok, prep.message
        prep.FDTD.NrTS = 2500
        r = s.run_prepared_hip(prep, frequency_hz=5.8e9, verbose=0)
        assert r.ok, r.message
        res.append(r)
    g, c = res
    assert g.intensity.shape == c.intensity.shape == (46, 25)
    assert rel_l2(g.s11, c.s11) < 1e-3 and rel_l2(g.port_u, c.port_u) < 1e-4 and rel_l2(g.port_i, c.port_i) < 1e-4
    _, ge, gh = sink.principal_cuts(g.theta, g.phi, g.intensity)
    _, ce, ch = sink.principal_cuts(c.theta, c.phi, c.intensity)
    lin = lambda d: 10.0 ** (np.asarray(d) / 20.0)
    assert rel_l2(lin(ge), lin(ce)) < 1e-3 and rel_l2(lin(gh), lin(ch)) < 1e-3
    assert abs(g.Dmax - c.Dmax) < 1e-6 * c.Dmax
    assert g.stats["steps"] == c.stats["steps"] and g.stats["grid"] == c.stats["grid"]
    data = sink.write_result(g, str(tmp_path / "result.json"))
    assert data["ok"] and len(data["e_plane_dBi"]) == 46


def test_multi_patch_rotated_runs_on_gpu(hip_lib, tmp_path):
    """Two elements, one rotated 90 deg about z and lifted, finite-thickness copper, volumetric lumped ports
    (multi_3d variant): runs on the GPU, both ports deliver series, pattern is finite."""
    s = pkg("solver_fdtd_hip")
    P = pkg("params").PatchAntennaParams
    p = P.from_user_units(frequency_ghz=5.8, er=4.3, h_mm=1.6, loss_tangent=0.02, metal_thickness_um=200.0)
    arr = [s.PatchInstance("A", p, -0.02, 0.0, 0.0, s.FeedDirection.NEG_X),
           s.PatchInstance("B", p, 0.02, 0.0, 0.002, s.FeedDirection.NEG_Y, rot_z_deg=90.0)]
    prep = s.prepare_hip_microstrip_multi_3d(arr, boundary="PML_8", theta_step_deg=6.0, phi_step_deg=30.0, mesh_quality=1,
                                             auto_margin_mm=(25.0, 25.0, 30.0), work_dir=str(tmp_path / "m"))
    assert prep.ok, prep.message
    prep.FDTD.NrTS = 1500
    r = s.run_prepared_hip(prep, frequency_hz=5.8e9, verbose=0)
    assert r.ok, r.message
    assert np.isfinite(r.intensity).all() and r.intensity.shape == (31, 13)
    series = prep.FDTD.sim.port_series()
    assert len(series) == 2 and all(np.abs(u).max() > 0 and np.abs(i).max() > 0 for u, i in series)


def test_northstar_acceptance(hip_lib, oracle_lib):
    """BASELINE north-star grid (300x300x60, CPML-10, lumped port, NF2FF surfaces) stepped on the HIP library and on
    the oracle: port series, S11(f), E/H-plane cuts and Dmax within the north-star's 1e-3 relative-L2 tolerance
    (they are in fact identical to rounding: same fp32 operation order).  3000 of the 12000 steps of the full
    acceptance run (tests/acceptance_northstar.py; its r01 result is committed under profiles/r01/)."""
    import acceptance_northstar
    out = acceptance_northstar.run(steps=3000, workload="NS")
    assert out["gpu"]["backend"].startswith("hip") and out["cpu_oracle"]["backend"].startswith("oracle")
    assert out["tolerance"] == 1e-3
    assert out["pass"], out["rel_l2"]
    assert max(out["rel_l2"].values()) < 1e-6, out["rel_l2"]


@pytest.mark.parametrize("overlap", ["split", "nosplit"])
def test_rccl_self_loopback_transport(hip_lib, oracle_lib, overlap):
    """The RCCL calls themselves on one GPU: an interior slab (rank 1 of 3) whose halos go to ITSELF through
    ncclSend/ncclRecv in a communicator of one (FDTD_FLAG_LOOPBACK) — the library's own step loop, communication
    stream, events and overlap split — against the same slab stepped by half-steps with its halo planes copied back
    through the host, on the HIP library and on the oracle.  Interior planes have live coefficients on both faces, so
    a wrong plane, component, count or ordering changes the fields."""
    capi = pkg("_capi")
    flag = capi.FLAG_LOOPBACK | (capi.FLAG_OVERLAP_ON if overlap == "split" else capi.FLAG_OVERLAP_OFF)
    n = 120

    def slab(lib, flags):
        s = patch_sim(44, 40, 36, nr_ts=n + 8, nf2ff=False)
        e = s.build(lib, rank=1, world=3, flags=flags)
        seeded_fields(e, 11)
        return s, e

    _, er = slab(hip_lib, flag)
    er.comm_init(capi.comm_unique_id(hip_lib))
    er.run(n)
    outs = [er.fields()]
    for lib in (hip_lib, oracle_lib):
        _, e = slab(lib, 0)
        for _ in range(n):
            e.half_step(capi.PHASE_E)
            e.halo_put(capi.HALO_E_DOWN, e.halo_get(capi.HALO_E_DOWN))
            e.half_step(capi.PHASE_H)
            e.halo_put(capi.HALO_H_UP, e.halo_get(capi.HALO_H_UP))
        outs.append(e.fields())
    assert er.step == n and np.isfinite(outs[2]).all() and np.abs(outs[2]).max() > 0
    assert same_values(outs[0], outs[1]), f"RCCL loopback vs host loopback (HIP): rel L2 {rel_l2(outs[0], outs[1]):.3e}"
    assert same_values(outs[0], outs[2]), f"RCCL loopback vs oracle: rel L2 {rel_l2(outs[0], outs[2]):.3e}"
    # and the loopback really matters: without any exchange the same slab ends elsewhere
    _, e0 = slab(hip_lib, 0)
    for _ in range(n):
        e0.half_step(capi.PHASE_E)
        e0.half_step(capi.PHASE_H)
    assert not same_values(e0.fields(), outs[0])


def _attach_p2p(engs):
    blobs = [e.p2p_export() for e in engs]
    for r, e in enumerate(engs):
        e.p2p_attach(blobs[r - 1] if r > 0 else None, blobs[r + 1] if r + 1 < len(engs) else None)


@pytest.mark.parametrize("world", [2, 3, 5])
def test_p2p_mailbox_transport_slabs_equal_one_slab(hip_lib, world):
    """P2P mailbox transport (halos pushed by the update kernels into the neighbour's mailbox, step-counter flags,
    dependent plane scheduled last): `world` slabs in this process on one GPU — each on its own stream, coupled only
    through the mailboxes — must reproduce the single-slab run bit for bit, over several fdtd_run_linked calls."""
    capi = pkg("_capi")
    s1 = patch_sim(56, 52, 34, nr_ts=260)
    e1 = s1.build(hip_lib)
    e1.run(260)
    sims = [patch_sim(56, 52, 34, nr_ts=260) for _ in range(world)]
    engs = [s.build(hip_lib, rank=r, world=world) for r, s in enumerate(sims)]
    _attach_p2p(engs)
    for n in (1, 100, 159):
        capi.run_linked(engs, n)
    f2 = np.concatenate([e.fields() for e in engs], axis=2)
    assert np.array_equal(e1.fields().view(np.uint32), f2.view(np.uint32))
    u1, i1 = s1.port_series()[0]
    u2 = sum(s.port_series()[0][0] for s in sims)
    i2 = sum(s.port_series()[0][1] for s in sims)
    assert rel_l2(u2, u1) < 1e-12 and rel_l2(i2, i1) < 1e-12
    for a, *parts in zip(s1.nf2ff_boxes(), *[s.nf2ff_boxes() for s in sims]):
        assert rel_l2(sum(parts), a) < 1e-12


def test_p2p_wait_times_out_instead_of_hanging(hip_lib):
    """A rank whose neighbour never steps: the bounded halo wait gives up (10 s) and fdtd_run reports it."""
    capi = pkg("_capi")
    sims = [patch_sim(40, 36, 24, nr_ts=20, nf2ff=False) for _ in range(2)]
    engs = [s.build(hip_lib, rank=r, world=2) for r, s in enumerate(sims)]
    _attach_p2p(engs)
    with pytest.raises(capi.FdtdError, match="timed out"):
        engs[0].run(1)          # rank 1 is never stepped: rank 0's H half-step waits for an E halo that does not come


@pytest.mark.parametrize("src_i", [1, 9])
def test_mur_with_sources_next_to_and_away_from_a_face(hip_lib, oracle_lib, src_i):
    """Mur scene driven through the C ABI directly: a source edge ON the plane next to the x- Mur face (src_i = 1: the
    order 'Mur post, then source' matters, so the library must take the unfused launch sequence) and in the interior
    (src_i = 9: sources/probes stay fused in the main kernels) — both identical to the oracle, fields and probe series."""
    capi, const = pkg("_capi"), pkg("constants")
    from opbuild_cases import random_scene
    grid, eps, kap, pec, _ = random_scene(21, (22, 20, 18), False, 3, n_lumped=0, pec_frac=0.0)
    dt = grid.courant_dt()
    nx, ny, nz = grid.shape
    n = 160
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
        e.set_signal(np.sin(2 * np.pi * 8e9 * t) * np.exp(-((t - 40 * dt) / (15 * dt)) ** 2))
        j, k = ny // 2, nz // 2
        src = np.array([(k * ny + j) * nx + src_i, (k * ny + j + 1) * nx + src_i], np.int64)
        e.add_source(src, np.array([2, 2], np.int8), np.array([1.0, -0.5], np.float32))
        pv = e.add_probe(capi.KIND_V, np.array([(k * ny + j) * nx + src_i + 3], np.int64), np.array([2], np.int8), np.array([-1.0], np.float32))
        pi = e.add_probe(capi.KIND_I, np.array([(k * ny + j) * nx + src_i + 3], np.int64), np.array([1], np.int8), np.array([1.0], np.float32))
        for m in (1, 70, n - 71):
            e.run(m)
        out.append((e.fields(), e.get_probe(pv), e.get_probe(pi)))
    (fh, uh, ih), (fo, uo, io) = out
    assert np.abs(fo).max() > 0 and len(uo) == n
    assert same_values(fh, fo), f"rel L2 {rel_l2(fh, fo):.3e}"
    assert rel_l2(uh, uo) < 1e-12 and rel_l2(ih, io) < 1e-12


def test_largest_baseline_grid_two_p2p_slabs_equal_one(hip_lib):
    """BASELINE's largest configuration (C5: 800x800x120, 2x2 array, four lumped ports, 76.8 Mcells) at full size:
    two z-slabs coupled by the P2P mailbox transport reproduce the single-slab run bit for bit — exercises the index
    arithmetic, the source lists and the mailboxes at the sizes the 8-GPU configuration is quoted on."""
    capi, sim_m, wl, sc = pkg("_capi"), pkg("simulation"), pkg("workloads"), pkg("scene")
    w = wl.baseline_workload("C5")
    vox = sc.voxelize(w.scene, w.grid)
    steps = 36

    def make():
        return sim_m.Simulation(w.grid, vox, f0=w.f0, fc=w.fc, boundary="CPML", cpml_cells=10, nr_ts=steps + 4, nf2ff_freqs=None)

    s1 = make()
    e1 = s1.build(hip_lib)
    e1.run(steps)
    ref = [e1.get_field(kind, comp) for kind in (0, 1) for comp in range(3)]
    u1 = [np.asarray(s1.port_series()[q][0]) for q in range(len(vox.ports))]
    del e1, s1
    sims = [make(), make()]
    engs = [s.build(hip_lib, rank=r, world=2) for r, s in enumerate(sims)]
    _attach_p2p(engs)
    capi.run_linked(engs, steps)
    n = 0
    for kind in (0, 1):
        for comp in range(3):
            both = np.concatenate([e.get_field(kind, comp) for e in engs], axis=0)
            assert both.shape == ref[n].shape
            assert np.array_equal(both.view(np.uint32), ref[n].view(np.uint32)), (kind, comp)
            n += 1
    assert max(np.abs(r).max() for r in ref) > 0
    for q in range(len(vox.ports)):
        u2 = sum(np.asarray(s.port_series()[q][0]) for s in sims)
        assert rel_l2(u2, u1[q]) < 1e-12


def test_c4_full_size_gpu_equals_oracle(hip_lib, oracle_lib):
    """BASELINE config 4 at full size (512x512x128, 5.8 GHz microstrip-3D geometry, CPML-10): 40 timesteps on the HIP
    library and on the oracle, every field value identical."""
    sim_m, wl, sc = pkg("simulation"), pkg("workloads"), pkg("scene")
    w = wl.baseline_workload("C4")
    vox = sc.voxelize(w.scene, w.grid)
    out = []
    for lib in (hip_lib, oracle_lib):
        s = sim_m.Simulation(w.grid, vox, f0=w.f0, fc=w.fc, boundary="CPML", cpml_cells=10, nr_ts=48, nf2ff_freqs=None)
        e = s.build(lib)
        e.run(40)
        out.append(([e.get_field(kind, comp) for kind in (0, 1) for comp in range(3)], np.asarray(s.port_series()[0][0])))
        del e, s
    (fh, uh), (fo, uo) = out
    assert max(np.abs(a).max() for a in fo) > 0
    for a, b in zip(fh, fo):
        assert same_values(a, b)
    assert rel_l2(uh, uo) < 1e-12

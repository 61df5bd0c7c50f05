"""Host-side logic (no GPU): mesher invariants, operator forms, slab partitioning, CPML slab tables,
rotated-box voxelisation, and the end-to-end plugin path with the oracle injected as the engine."""
import numpy as np
import pytest

from conftest import pkg
from helpers import patch_sim


def test_smooth_mesh_lines_invariants():
    m = pkg("mesher")
    rng = np.random.default_rng(0)
    for _ in range(20):
        hints = np.concatenate([[-100.0, 100.0], rng.uniform(-60, 60, 8), [0.0, 0.4, 0.8, 1.2, 1.6]])
        out = m.smooth_mesh_lines(hints, 4.08, 1.4)
        d = np.diff(out)
        assert np.all(d > 0) and d.max() <= 4.08 * (1 + 1e-9)
        for h in hints:                                     # every hint line survives — or, where two random hints fell closer than
            assert np.min(np.abs(out - h)) < 4.08 / 100     # max_res / 100, as ONE line at their mean (merge_close_lines)
        # grading: away from forced hint lines the neighbour ratio stays within `ratio`
        ratio = np.maximum(d[1:] / d[:-1], d[:-1] / d[1:])
        forced = np.array([np.min(np.abs(hints - x)) < 4.08 / 100 for x in out])
        free = ~(forced[1:-1])
        assert np.all(ratio[free] <= 1.4 * 1.02)


def test_excitation_longer_than_the_run_is_flagged():
    import warnings
    wl, sc, simm = pkg("workloads"), pkg("scene"), pkg("simulation")
    w = wl.patch_workload("t", nx=40, ny=40, nz=30)
    vox = sc.voxelize(w.scene, w.grid)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        s = simm.Simulation(w.grid, vox, f0=w.f0, fc=w.fc, boundary="PEC", nr_ts=50)
        assert s.excitation_warning and "NrTS = 50" in s.excitation_warning and any("ends before the pulse" in str(c.message) for c in caught)
        assert simm.Simulation(w.grid, vox, f0=w.f0, fc=w.fc, boundary="PEC", nr_ts=len(s.signal)).excitation_warning is None


def test_hint_lines_micrometres_apart_become_one():
    """Independent hint sets landing 6.8 um apart (the reference's 2 x 2 array at a 61.2 mm pitch, y axis) on a 3.4 mm mesh: one line,
    so that the Courant timestep is set by the mesh and not by the accident; the outermost lines and everything else stay."""
    m = pkg("mesher")
    hints = [-80.0, -31.9245, -31.9177, -10.0, 0.0, 0.4, 0.8, 29.2755, 29.2823, 80.0]
    out = m.smooth_mesh_lines(hints, 3.4, 1.4)
    assert np.diff(out).min() >= 0.034 and out[0] == -80.0 and out[-1] == 80.0
    assert np.min(np.abs(out - (-31.9211))) < 1e-9 and np.min(np.abs(out - 29.2789)) < 1e-9
    for h in (-10.0, 0.0, 0.4, 0.8):
        assert np.min(np.abs(out - h)) < 1e-12
    assert np.array_equal(m.merge_close_lines(np.array([0.0, 0.001, 1.0, 1.9995, 2.0]), 0.01), [0.0, 1.0, 2.0])   # the ends do not move
    # lines that are close ON PURPOSE keep their equally close neighbours: five lines across a 0.254 mm substrate under an 11 mm mesh
    sub = [-150.0, -40.0, 0.0, 0.0635, 0.127, 0.1905, 0.254, 40.0, 150.0]
    out = m.smooth_mesh_lines(sub, 11.0, 1.4)
    for h in sub:
        assert np.min(np.abs(out - h)) < 1e-12


def test_thirds_rule_hint():
    m = pkg("mesher")
    h = m.mesh_hint_from_box([-18.0, -14.0, 1.6], [18.0, 14.0, 1.6], [0, 1], metal_edge_res=2.1)
    assert h[2] is None
    assert sorted(h[0]) == pytest.approx(sorted([-18 + 0.7, -18 - 1.4, 18 - 0.7, 18 + 1.4]))
    assert m.mesh_hint_from_box([-6, 0, 0], [-6, 0, 1.6], [0, 1], None)[:2] == [[-6.0], [0.0]]


def test_slab_range_partitions():
    sr = pkg("simulation").slab_range
    for nz in (40, 60, 61, 120, 128):
        for w in (1, 2, 3, 4, 8):
            parts = [sr(nz, w, r) for r in range(w)]
            assert parts[0][0] == 0 and sum(nk for _, nk in parts) == nz
            for (a, n), (b, _) in zip(parts[:-1], parts[1:]):
                assert a + n == b
            assert max(nk for _, nk in parts) - min(nk for _, nk in parts) <= 1


def test_cost_weighted_slab_partition():
    """simulation.slab_partition: contiguous, complete, >= 2 planes per slab, never worse than the even split in its own
    cost model, and within 8 % of perfect balance on the BASELINE decompositions (C5 over 8, C4 over 4)."""
    sim = pkg("simulation")
    for nz, world, cl in ((60, 8, 10), (60, 4, 10), (60, 2, 10), (128, 4, 10), (120, 8, 10), (40, 2, 10), (16, 5, 3), (36, 3, 8)):
        cost = sim.plane_costs(nz, cl, cl)
        assert cost.size == nz and (cost > 1).sum() == 2 * cl + 1           # the top layer runs through the last (inert) line
        parts = sim.slab_partition(cost, world)
        assert parts[0][0] == 0 and sum(nk for _, nk in parts) == nz and all(nk >= 2 for _, nk in parts)
        for (a, n), (b, _) in zip(parts[:-1], parts[1:]):
            assert a + n == b
        mine = [cost[a:a + n].sum() for a, n in parts]
        even = [cost[a:a + n].sum() for a, n in (sim.slab_range(nz, world, r) for r in range(world))]
        assert max(mine) <= max(even) + 1e-12
        assert parts == sim.slab_partition(cost, world)                          # deterministic: every rank computes it alone
        if (nz, world) in ((120, 8), (128, 4)):
            assert max(mine) / np.mean(mine) <= 1.03 and max(even) / np.mean(even) > 1.06
    with pytest.raises(ValueError):
        sim.slab_partition(np.ones(7), 4)
    # Simulation.slabs: cost-weighted by default when there are z layers, even otherwise / on request
    s = patch_sim(40, 40, 60, cpml_cells=10, nr_ts=10, nf2ff=False)
    assert s.slabs(1) == [(0, 60)] and s.slabs(4, "even") == [sim.slab_range(60, 4, r) for r in range(4)]
    assert [nk for _, nk in s.slabs(4)] == [13, 17, 17, 13]
    assert patch_sim(40, 40, 60, boundary="PEC", nr_ts=10, nf2ff=False).slabs(4) == s.slabs(4, "even")


def test_run_cleanup_wipes_a_stale_sim_path(oracle_lib, tmp_path, monkeypatch):
    """openems_api.Run(sim_path, cleanup=True) removes a pre-existing sim_path first, as the reference's engine does
    (fixed.py:280) — but never the directory the process runs in; the port text files are only written on request."""
    import os
    import tutorial_scene
    stale = tmp_path / "run" / "old_dump.h5"
    stale.parent.mkdir()
    stale.write_text("x")
    r = tutorial_scene.build_and_run(oracle_lib, str(tmp_path / "run"), nr_ts=40)
    assert r["steps"] == 40 and not stale.exists() and os.path.isfile(tmp_path / "run" / "fdtd_hip_run.json")
    assert not os.path.exists(tmp_path / "run" / "port_ut1")
    keep = tmp_path / "cwd" / "precious.txt"
    keep.parent.mkdir()
    keep.write_text("x")
    monkeypatch.chdir(keep.parent)
    monkeypatch.setenv("FDTD_WRITE_PORT_FILES", "1")
    tutorial_scene.build_and_run(oracle_lib, str(keep.parent), nr_ts=40)
    assert keep.exists() and os.path.isfile(keep.parent / "port_ut1") and os.path.isfile(keep.parent / "port_it1")


def test_cpml_slab_tables_are_end_anchored():
    s = patch_sim(40, 40, 60, cpml_cells=10, nr_ts=10, nf2ff=False)
    n_full = None
    for world in (1, 2, 4, 8):
        tot = 0
        for r in range(world):
            k0, nk = pkg("simulation").slab_range(60, world, r)
            sx, sy, sz, nsx, nsy, nsz, coef = s.cpml.for_slab(k0, nk)
            assert coef.size == 6 * (40 + 40 + nk)
            own = sz[sz >= 0]
            assert np.array_equal(own, np.arange(nsz))         # compact local numbering
            lo = 0
            while lo < nk and sz[lo] == lo:
                lo += 1
            hi = lo
            while hi < nk and sz[hi] < 0:
                hi += 1
            assert np.array_equal(sz[hi:], lo + np.arange(nk - hi))  # [0,lo) and [hi,nk) — what the kernels assume
            tot += nsz
        n_full = tot if n_full is None else n_full
        assert tot == n_full


def test_operator_class_and_raw_forms_agree(oracle_lib):
    """The compressed operator (class bytes + 1-D tables) expands to exactly the raw arrays."""
    s = patch_sim(36, 34, 30, nr_ts=60, nf2ff=False)
    cls = s.op.classes()
    assert cls is not None and cls[1].size <= 256
    res = []
    for use in (True, False):
        s2 = patch_sim(36, 34, 30, nr_ts=60, nf2ff=False, use_classes=use)
        e = s2.build(oracle_lib)
        e.run(60)
        res.append(e.fields())
        assert s2.operator_form == ("classes" if use else "raw")
    assert np.array_equal(res[0], res[1])


def test_rotated_box_voxelisation():
    sc, g = pkg("scene"), pkg("grid")
    grid = g.RectGrid(np.linspace(-20e-3, 20e-3, 41), np.linspace(-20e-3, 20e-3, 41), np.linspace(-10e-3, 10e-3, 21))
    M = np.eye(4)
    ang = np.deg2rad(90.0)
    M[:3, :3] = [[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]]
    M[:3, 3] = [2.0, 0.0, 1.0]
    a, b = sc.Scene(), sc.Scene()
    a.add_metal("m").boxes.append(sc.Box((-10.0, -4.0, 0.0), (10.0, 4.0, 0.0), 10, M))
    b.add_metal("m").add_box([-4.0 + 2.0, -10.0, 1.0], [4.0 + 2.0, 10.0, 1.0], 10)   # the same sheet, axis aligned
    va, vb = sc.voxelize(a, grid), sc.voxelize(b, grid)
    assert va.pec.sum() > 0 and np.array_equal(va.pec, vb.pec)
    a.add_material("d", 4.3, 0.01).boxes.append(sc.Box((-10.0, -4.0, -1.0), (10.0, 4.0, 0.0), 0, M))
    b.add_material("d", 4.3, 0.01).add_box([-2.0, -10.0, 0.0], [6.0, 10.0, 1.0], 0)
    va, vb = sc.voxelize(a, grid), sc.voxelize(b, grid)
    assert (va.eps_r > 1).sum() > 0 and np.array_equal(va.eps_r, vb.eps_r)
    # a 30-degree sheet is a staircase with about the same area
    ang = np.deg2rad(30.0)
    M2 = np.eye(4)
    M2[:3, :3] = [[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]]
    c = sc.Scene()
    c.add_metal("m").boxes.append(sc.Box((-10.0, -4.0, 0.0), (10.0, 4.0, 0.0), 10, M2))
    vc = sc.voxelize(c, grid)
    ref = sc.voxelize(b, grid).pec.sum()
    assert 0.8 * ref < vc.pec.sum() < 1.2 * ref


def test_fixed_scene_end_to_end_with_oracle_engine(oracle_lib, tmp_path):
    """PatchAntennaParams in -> OpenEMSResult-shaped object out, through the plugin surface, with the
    oracle standing in for the GPU library (test hook `lib`).  Physics bands:
    this variant puts W = 37.6 mm on the resonant axis, so the S11 dip sits near 1.9 GHz."""
    s = pkg("solver_fdtd_hip")
    P = pkg("params").PatchAntennaParams
    p = P.from_user_units(frequency_ghz=2.45, er=4.3, h_mm=1.6, loss_tangent=0.02)
    prep = s.prepare_hip_patch_fixed(p, work_dir=str(tmp_path / "run"), lib=oracle_lib)
    assert prep.ok, prep.message
    prep.FDTD.NrTS = 9000
    r = s.run_prepared_hip(prep, frequency_hz=2.45e9, verbose=0)
    assert r.ok, r.message
    assert r.is_dBi and r.intensity.shape == (90, 2)
    assert r.theta.shape == (90,) and abs(r.theta[1] - np.deg2rad(2.0)) < 1e-12 and np.allclose(r.phi, [0, np.pi / 2])
    assert abs(r.intensity.max() - 10 * np.log10(r.Dmax)) < 1e-9
    assert 4.0 < 10 * np.log10(r.Dmax) < 9.0                     # patch-antenna directivity band
    assert np.argmax(r.intensity[:, 0]) < 20                      # main lobe near broadside
    k = int(np.argmin(r.s11_dB))
    assert 1.75e9 < r.freq[k] < 2.05e9 and r.s11_dB[k] < -6.0
    assert r.stats["cells"] > 1e5 and r.stats["steps"] == 9000
    import os
    assert os.path.isfile(os.path.join(r.sim_path, "fdtd_hip_run.json"))


def test_backward_flag_set_covers_every_block_an_e_block_reads():
    """Several timesteps per launch: an E block of timestep s + 1 waits for the H flags of `wf_wait_back`'s block set
    (csrc/kernel_common.hpp).  Brute force over tilings — rows shorter and longer than a block, short last strips, one-row
    strips: every thread whose I values a block's threads read (own cell group, the row below, the group to the left)
    lies in a block of that set.  (The plane below is the same block index, trivially.)  The arithmetic below restates the
    device code's; the geometry it is checked against is enumerated cell by cell."""
    B = 256
    for P4, ny, tys in ((14, 52, 13), (75, 300, 17), (50, 200, 40), (257, 9, 4), (600, 12, 5), (16, 9, 9), (100, 400, 40), (9, 60, 1)):
        nstrips = (ny + tys - 1) // tys
        nbs = (min(tys, ny) * P4 + B - 1) // B
        hop = 1 + P4 // B
        for strip in range(nstrips):
            rows = min(tys, ny - strip * tys)
            T = rows * P4
            for pb in range(nbs):
                lo, hi = pb * B, min((pb + 1) * B, T)
                if lo >= T:
                    continue
                # what the device waits for: (strip delta, block) pairs
                first = pb * B - P4 - 1
                dev = set()
                for t in range(hop + 1):
                    if pb - t >= 0 and (pb - t + 1) * B - 1 >= first:
                        dev.add((0, pb - t))
                Tp = tys * P4
                lastb = (Tp - 1) // B
                for q in range(hop + 1):
                    if first < 0 and strip > 0 and lastb - q >= 0 and min((lastb - q + 1) * B, Tp) - 1 >= Tp + first:
                        dev.add((-1, lastb - q))
                # what the block's threads read, thread by thread
                need = set()
                for t in range(lo, hi):
                    jj, c = divmod(t, P4)
                    need.add((0, t // B))                                   # own cells
                    if c > 0:
                        need.add((0, (t - 1) // B))                         # the element left of the group: thread t - 1, same row
                    if jj > 0:
                        need.add((0, (t - P4) // B))                        # row below, same strip
                    elif strip > 0:
                        need.add((-1, (Tp - P4 + c) // B))                  # row below = last row of the previous (full) strip
                assert need <= dev, (P4, ny, tys, strip, pb, sorted(need - dev))
                assert len(dev) <= 2 * hop + 2                              # fits the polling lanes (one more lane: plane k - 1)


def test_nf2ff_box_on_a_metal_sheet_is_flagged():
    """A Huygens face that lies ON the ground plane (BASELINE config 2's 40 planes with a third of the spare ones below the ground plane:
    plane 12 = layer + 2) gives a far field that is not the antenna's (D read 14.6 dBi for a 6.2 dBi patch): Simulation warns and keeps
    the text; the shipped C2 workload puts the ground plane on plane 14 and is clean, like the other BASELINE grids."""
    import warnings
    wl, sc, simm = pkg("workloads"), pkg("scene"), pkg("simulation")

    def build(w):
        vox = sc.voxelize(w.scene, w.grid)
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            s = simm.Simulation(w.grid, vox, f0=w.f0, fc=w.fc, boundary="CPML", cpml_cells=10, nr_ts=100000, nf2ff_freqs=[w.f0])
        return s, [c for c in caught if issubclass(c.category, RuntimeWarning)]
    s, caught = build(wl.patch_workload("third-below", nx=100, ny=100, nz=40))
    assert s.nf2ff_warning and "z-min face (node plane 12)" in s.nf2ff_warning and len(caught) == 1
    for name in ("C2", "NS"):
        s, caught = build(wl.baseline_workload(name))
        assert s.nf2ff_warning is None and not caught


def test_mesher_merge_is_optional_and_reported():
    """merge_close_lines is a deliberate deviation from the toolkit the reference calls (which keeps every hint line): merge_fraction = 0
    reproduces its lines, and what was merged is reported so that a user can see that the mesh differs."""
    m = pkg("mesher")
    lines = [0.0, 10.0, 10.0000068, 20.0, 30.0]
    rec = []
    merged = m.smooth_mesh_lines(lines, 3.4, 1.4, merged=rec)
    assert rec == [(10.0, 10.0000068)] and np.min(np.diff(merged)) > 1.0
    kept = m.smooth_mesh_lines(lines, 3.4, 1.4, merge_fraction=0)
    assert np.isclose(np.min(np.diff(kept)), 6.8e-6) and all(np.any(np.isclose(kept, v, rtol=0, atol=1e-12)) for v in lines)
    # through the mirrored mesh object: the pairs per axis
    oa = pkg("openems_api")
    g = oa.ContinuousStructure().GetGrid()
    g.AddLine("y", lines)
    g.SmoothMeshLines("y", 3.4, 1.4)
    assert g.merged_lines[1] == [(10.0, 10.0000068)] and not g.merged_lines[0]
    g2 = oa.ContinuousStructure().GetGrid()
    g2.merge_fraction = 0
    g2.AddLine("y", lines)
    g2.SmoothMeshLines("y", 3.4, 1.4)
    assert not g2.merged_lines[1] and np.isclose(np.min(np.diff(np.sort(g2.GetLines("y")))), 6.8e-6)

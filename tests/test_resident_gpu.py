"""The resident schedule (csrc/resident.hip, FDTD_FLAG_KERNEL_RESIDENT): small grids without CPML layers stepped with the whole grid
in registers, tile halos as data-tagged granules.  HIP vs the oracle through the C ABI, fields bit for bit; and against the
two-launch schedule of the HIP library itself (probe series and NF2FF spectra identical: same reduction trees)."""
import numpy as np
import pytest

from conftest import pkg
from helpers import patch_sim, seeded_fields, rel_l2

pytestmark = pytest.mark.gpu


def _fields_equal(fh, fo):
    assert np.isfinite(fo).all() and np.abs(fo).max() > 0
    assert np.array_equal(fh, fo), f"rel L2 {rel_l2(fh, fo):.3e}"
    nz = fo != 0
    assert np.array_equal(fh[nz].view(np.uint32), fo[nz].view(np.uint32))


SHAPES = [
    (56, 55, 50),    # the reference GUI's default grid: plane pairs, 7 strips of 7-8 rows
    (53, 47, 31),    # nx not a multiple of 4 (x-high face inside a group), odd plane count (one single-plane tile)
    (49, 44, 30),    # nx - 1 a multiple of 4: the x-high boundary cell is the FIRST of its group, its inner cell in the thread before
    (130, 21, 12),   # rows of 33 groups: tiles of 2 planes x 3 rows
    (260, 23, 17),   # rows of 65 groups: single-plane tiles of 3 rows (no Mur z faces possible)
]


@pytest.mark.parametrize("use_classes", [True, False])
@pytest.mark.parametrize("shape", SHAPES[:3])
def test_resident_mur_fields_equal_the_oracle(hip_lib, oracle_lib, shape, use_classes):
    """Mur on all six faces, seeded fields (every stencil term and every Mur face live from the first timestep), source and probes."""
    capi = pkg("_capi")
    res = []
    for lib, flags in ((hip_lib, capi.FLAG_KERNEL_RESIDENT), (oracle_lib, 0)):
        s = patch_sim(*shape, boundary="MUR", nr_ts=300, use_classes=use_classes)
        e = s.build(lib, flags=flags)
        seeded_fields(e, 11)
        e.run(300)
        res.append((s, e))
    (sh, eh), (so, eo) = res
    info = eh.schedule_info()
    assert info["resident"] and info["launches_per_timestep"] == 1
    _fields_equal(eh.fields(), eo.fields())
    (uh, ih), (uo, io) = sh.port_series()[0], so.port_series()[0]
    assert rel_l2(uh, uo) < 1e-12 and rel_l2(ih, io) < 1e-12
    for a, b in zip(sh.nf2ff_boxes(), so.nf2ff_boxes()):
        assert rel_l2(a, b) < 1e-12


@pytest.mark.parametrize("bc", [["PEC"] * 6, ["MUR", "PEC", "PEC", "MUR", "MUR", "PEC"], ["PEC", "MUR", "MUR", "PEC", "PEC", "MUR"]])
def test_resident_pec_and_mixed_faces(hip_lib, oracle_lib, bc):
    capi = pkg("_capi")
    res = []
    for lib, flags in ((hip_lib, capi.FLAG_KERNEL_RESIDENT), (oracle_lib, 0)):
        s = patch_sim(50, 46, 33, boundary=bc, nr_ts=250)
        e = s.build(lib, flags=flags)
        seeded_fields(e, 5)
        e.run(250)
        res.append(e)
    assert res[0].schedule_info()["resident"]
    _fields_equal(res[0].fields(), res[1].fields())


@pytest.mark.parametrize("shape,bc", [((130, 21, 12), "MUR"), ((260, 23, 17), ["MUR", "MUR", "MUR", "MUR", "PEC", "PEC"])])
def test_resident_wide_rows(hip_lib, oracle_lib, shape, bc):
    """Rows of 33 and 65 four-cell groups: few rows per tile; 65 groups leave no room for plane pairs (then no Mur z faces)."""
    capi = pkg("_capi")
    res = []
    for lib, flags in ((hip_lib, capi.FLAG_KERNEL_RESIDENT), (oracle_lib, 0)):
        s = patch_sim(*shape, boundary=bc, nr_ts=200, nf2ff=False)
        e = s.build(lib, flags=flags)
        seeded_fields(e, 7)
        e.run(200)
        res.append(e)
    assert res[0].schedule_info()["resident"]
    _fields_equal(res[0].fields(), res[1].fields())


@pytest.mark.parametrize("use_classes", [True, False])
@pytest.mark.parametrize("shape,cells", [((56, 55, 50), 8), ((53, 47, 31), 6), ((49, 44, 30), 3), ((130, 21, 26), 5)])
def test_resident_cpml_fields_equal_the_oracle(hip_lib, oracle_lib, shape, cells, use_classes):
    """CPML on all six faces inside the resident kernel (psi values in registers): layers thinner than / not a multiple of the four-cell
    groups, odd plane counts, class and raw operator; port series and NF2FF spectra as well."""
    capi = pkg("_capi")
    res = []
    for lib, flags in ((hip_lib, capi.FLAG_KERNEL_RESIDENT), (oracle_lib, 0)):
        s = patch_sim(*shape, boundary="CPML", cpml_cells=cells, nr_ts=300, use_classes=use_classes)
        e = s.build(lib, flags=flags)
        seeded_fields(e, 13)
        e.run(300)
        res.append((s, e))
    (sh, eh), (so, eo) = res
    assert eh.schedule_info()["resident"]
    _fields_equal(eh.fields(), eo.fields())
    (uh, ih), (uo, io) = sh.port_series()[0], so.port_series()[0]
    assert rel_l2(uh, uo) < 1e-12 and rel_l2(ih, io) < 1e-12
    for a, b in zip(sh.nf2ff_boxes(), so.nf2ff_boxes()):
        assert rel_l2(a, b) < 1e-12


def test_resident_mixed_mur_cpml_pec_faces(hip_lib, oracle_lib):
    capi = pkg("_capi")
    bc = ["MUR", "CPML", "CPML", "MUR", "PEC", "CPML"]
    res = []
    for lib, flags in ((hip_lib, capi.FLAG_KERNEL_RESIDENT), (oracle_lib, 0)):
        s = patch_sim(50, 46, 33, boundary=bc, cpml_cells=6, nr_ts=300)
        e = s.build(lib, flags=flags)
        seeded_fields(e, 3)
        e.run(300)
        res.append(e)
    assert res[0].schedule_info()["resident"]
    _fields_equal(res[0].fields(), res[1].fields())


def test_resident_cpml_in_chunks_equals_two_launches(hip_lib):
    """psi, fields, probes and the NF2FF record over several resident launches (calls of 150 + 1 + 149 timesteps) ≡ 300 timesteps under two
    launches per timestep: identical bits."""
    capi = pkg("_capi")
    s1 = patch_sim(56, 55, 50, boundary="CPML", cpml_cells=8, nr_ts=300, nf2ff_mode="record")
    e1 = s1.build(hip_lib, flags=capi.FLAG_KERNEL_DIRECT)
    e1.run(300)
    s3 = patch_sim(56, 55, 50, boundary="CPML", cpml_cells=8, nr_ts=300, nf2ff_mode="record")
    e3 = s3.build(hip_lib, flags=capi.FLAG_KERNEL_RESIDENT)
    for n in (150, 1, 149):
        e3.run(n)
    assert e3.schedule_info()["resident"] and not e1.schedule_info()["resident"]
    assert np.array_equal(e1.fields(), e3.fields())
    for (u1, i1), (u3, i3) in zip(s1.port_series(), s3.port_series()):
        assert np.array_equal(u1, u3) and np.array_equal(i1, i3) and np.abs(u1).max() > 0
    for a, b in zip(s1.nf2ff_boxes(), s3.nf2ff_boxes()):
        assert np.array_equal(a, b)


def test_resident_equals_two_launch_schedule_in_chunks(hip_lib):
    """The same run under the resident schedule in calls of 1 / 7 / 100 / 61 / 131 timesteps and under three launches per timestep in one
    call: fields, probe series and recorded NF2FF spectra identical (the resident kernel cuts its launches at the sampled timesteps;
    probe sums by probe_block's tree)."""
    capi = pkg("_capi")
    out = []
    for flags, calls in ((capi.FLAG_KERNEL_RESIDENT, (1, 7, 100, 61, 131)), (capi.FLAG_KERNEL_DIRECT, (300,))):
        s = patch_sim(56, 55, 50, boundary="MUR", nr_ts=300, nf2ff_mode="record")
        e = s.build(hip_lib, flags=flags)
        for n in calls:
            e.run(n)
        out.append((s, e))
    (sr, er), (sd, ed) = out
    assert er.schedule_info()["resident"] and not ed.schedule_info()["resident"] and ed.schedule_info()["launches_per_timestep"] == 2
    assert er.step == ed.step == 300
    assert np.array_equal(er.fields(), ed.fields())
    for (ur, ir), (ud, id_) in zip(sr.port_series(), sd.port_series()):
        assert np.array_equal(ur, ud) and np.array_equal(ir, id_) and np.abs(ud).max() > 0
    for a, b in zip(sr.nf2ff_boxes(), sd.nf2ff_boxes()):
        assert np.array_equal(a, b)


def test_auto_takes_the_resident_schedule_for_mur_scenes_only(hip_lib):
    capi = pkg("_capi")
    assert patch_sim(56, 55, 50, boundary="MUR", nr_ts=10, nf2ff=False).build(hip_lib).schedule_info()["resident"]
    assert not patch_sim(56, 55, 50, boundary="MUR", nr_ts=10, nf2ff=False).build(hip_lib, flags=capi.FLAG_KERNEL_DIRECT).schedule_info()["resident"]
    # asked for by name where it cannot run: an error code, not a fallback
    e = patch_sim(300, 300, 60, boundary="MUR", nr_ts=10, nf2ff=False).build(hip_lib, flags=capi.FLAG_KERNEL_RESIDENT)
    with pytest.raises(capi.FdtdError, match="resident schedule"):
        e.run(2)
    # too many tiles for the chip: AUTO takes the one-launch schedule (k_step<..., MUR>: from 1700 blocks per sweep), below that two launches
    big = patch_sim(300, 300, 60, boundary="MUR", nr_ts=10, nf2ff=False).build(hip_lib)
    assert not big.schedule_info()["resident"] and big.schedule_info()["launches_per_timestep"] == 1
    mid = patch_sim(150, 140, 36, boundary="MUR", nr_ts=10, nf2ff=False).build(hip_lib)
    assert not mid.schedule_info()["resident"] and mid.schedule_info()["launches_per_timestep"] == 2


def test_resident_halo_timeout_heals_itself(hip_lib, monkeypatch):
    """Fault injection ($FDTD_WF_FAULT_STEP: the pulls of that launch wait for tags nobody publishes, 20 us): the C ABI reports an
    error instead of hanging, clears it, and Simulation.run repeats the run under the two-launch schedule."""
    capi = pkg("_capi")
    monkeypatch.setenv("FDTD_WF_FAULT_STEP", "20")
    s = patch_sim(56, 55, 50, boundary="MUR", nr_ts=200, nf2ff=False)
    e = s.build(hip_lib)
    monkeypatch.delenv("FDTD_WF_FAULT_STEP")
    with pytest.raises(capi.FdtdError, match="resident schedule"):
        e.run(100)
    s2 = patch_sim(56, 55, 50, boundary="MUR", nr_ts=200, nf2ff=False)
    monkeypatch.setenv("FDTD_WF_FAULT_STEP", "20")
    s2.build(hip_lib)
    monkeypatch.delenv("FDTD_WF_FAULT_STEP")
    st = s2.run(check_every=100, log=lambda *_: None)
    assert st.steps == 200 and st.schedule_fallback and "resident schedule" in st.schedule_fallback
    ref = patch_sim(56, 55, 50, boundary="MUR", nr_ts=200, nf2ff=False)
    er = ref.build(hip_lib, flags=capi.FLAG_KERNEL_DIRECT)
    er.run(200)
    assert np.array_equal(s2.engine.fields(), er.fields())

"""fdtd_build_operator on the GPU (csrc/opbuild.hip) against the oracle's C restatement (itself pinned to the numpy
spec in test_operator_build_cpu.py): expanded coefficients identical to the bit in all three device forms
(packed classes, per-edge classes, raw), on z-slabs, and — end to end — the same fields as the host set-up path."""
import numpy as np
import pytest

from conftest import pkg
from helpers import patch_sim
from opbuild_cases import random_scene, engine_with_built_operator, same_bits

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed,shape,graded,nmat,expect", [
    (1, (37, 30, 22), False, 3, "classes"),              # few materials, uniform mesh: <= 256 pairs (random PEC edges: > 256 triples)
    (4, (34, 29, 21), False, 8, None),                    # 8 materials: more pairs; packed or per-edge, whatever fits
    (2, (33, 27, 19), True, 8, "raw"),                    # graded mesh x 8 materials: > 256 pairs
])
@pytest.mark.parametrize("world,rank", [(1, 0), (3, 0), (3, 1), (3, 2)])
def test_hip_build_matches_oracle_build(hip_lib, oracle_lib, seed, shape, graded, nmat, expect, world, rank):
    grid, eps, kap, pec, lumped = random_scene(seed, shape, graded, nmat)
    dt = grid.courant_dt()
    eh, k0, nk = engine_with_built_operator(hip_lib, grid, eps, kap, pec, lumped, dt, rank=rank, world=world)
    eo, _, _ = engine_with_built_operator(oracle_lib, grid, eps, kap, pec, lumped, dt, rank=rank, world=world)
    assert eh.backend.startswith("hip") and eo.backend.startswith("oracle")
    for name, a, b in zip(("vv", "vi", "ii", "iv"), eh.get_operator(), eo.get_operator()):
        assert same_bits(a, b), name
    form_h, ncls_h = eh.operator_form()
    form_o, ncls_o = eo.operator_form()
    assert ncls_h == ncls_o and (form_h == "raw") == (form_o == "raw")
    if expect is not None and world == 1:
        assert form_h.startswith(expect)


def test_hip_build_forced_raw_and_per_edge_forms(hip_lib, oracle_lib):
    grid, eps, kap, pec, lumped = random_scene(1, (30, 26, 18), False, 3)
    dt = grid.courant_dt()
    eo, _, _ = engine_with_built_operator(oracle_lib, grid, eps, kap, pec, lumped, dt)
    er, _, _ = engine_with_built_operator(hip_lib, grid, eps, kap, pec, lumped, dt, prefer_classes=False)
    assert er.operator_form() == ("raw", 0)
    for a, b in zip(er.get_operator(), eo.get_operator()):
        assert same_bits(a, b)
    # many distinct triples but few pairs -> one byte per EDGE: uncorrelated single-cell materials, 6 of them
    rng = np.random.default_rng(0)
    nx, ny, nz = grid.shape
    pal_e, pal_k = np.array([1.0, 2.2, 4.3, 9.8, 6.15, 3.38]), np.array([0.0, 1e-3, 2e-3, 0.0, 5e-3, 0.1])
    mat = rng.integers(0, 6, size=(nz - 1, ny - 1, nx - 1))
    mat[:, :, : (nx - 1) // 2] = 0           # half the box uniform, so that pairs stay <= 256 ... if they do
    eps2, kap2 = pal_e[mat], pal_k[mat]
    ee, _, _ = engine_with_built_operator(hip_lib, grid, eps2, kap2, pec, lumped, dt)
    eo2, _, _ = engine_with_built_operator(oracle_lib, grid, eps2, kap2, pec, lumped, dt)
    for a, b in zip(ee.get_operator(), eo2.get_operator()):
        assert same_bits(a, b)
    assert ee.operator_form()[1] == eo2.operator_form()[1]


@pytest.mark.parametrize("use_classes", [True, False])
def test_fields_same_on_device_and_host_set_up(hip_lib, oracle_lib, use_classes):
    """Simulation.build with the device operator build (product default) vs numpy arrays uploaded, vs the oracle."""
    sims = [patch_sim(52, 44, 30, nr_ts=300, use_classes=use_classes) for _ in range(3)]
    sims[1].device_operator = False
    eh, eh_host, eo = sims[0].build(hip_lib), sims[1].build(hip_lib), sims[2].build(oracle_lib)
    for a, b, c in zip(eh.get_operator(), eh_host.get_operator(), eo.get_operator()):
        assert same_bits(a, b) and same_bits(a, c)
    for e in (eh, eh_host, eo):
        e.run(300)
    fh, fhh, fo = eh.fields(), eh_host.fields(), eo.fields()
    assert np.array_equal(fh, fhh) and np.array_equal(fh, fo) and np.abs(fo).max() > 0
    assert sims[0].operator_form == ("classes" if use_classes else "raw")
    if use_classes:          # a voxelised patch scene has a handful of classes: one byte per CELL on the device
        assert eh.operator_form()[0] == "classes-packed" and eh_host.operator_form()[0] == "classes-packed"
        assert eh.operator_form()[1] == eh_host.operator_form()[1] == eo.operator_form()[1]


@pytest.mark.parametrize("schedule", ["two_launches", "one_launch"])
def test_stepping_in_every_operator_form_equals_the_oracle(hip_lib, oracle_lib, schedule):
    """The update kernels with ONE CLASS BYTE PER EDGE (scenes with more than 256 distinct class triples: random PEC edges,
    uncorrelated single-cell materials) and with raw coefficient arrays, stepped 60 timesteps from random fields under both
    kernel schedules: fields identical to the oracle's (which steps the expanded coefficients).  The packed one-byte-per-cell
    form is what every patch scene of the suite runs in."""
    capi, eco, simm, const = pkg("_capi"), pkg("ecoperator"), pkg("simulation"), pkg("constants")
    flags = capi.FLAG_KERNEL_DIRECT if schedule == "two_launches" else capi.FLAG_KERNEL_WAVEFRONT
    grid, eps, kap, pec, lumped = random_scene(1, (30, 26, 18), False, 3)
    dt = grid.courant_dt()
    nx, ny, nz = grid.shape
    rng = np.random.default_rng(0)
    pal_e, pal_k = np.array([1.0, 2.2, 4.3, 9.8, 6.15, 3.38]), np.array([0.0, 1e-3, 2e-3, 0.0, 5e-3, 0.1])
    mat = rng.integers(0, 6, size=(nz - 1, ny - 1, nx - 1))
    mat[:, :, : (nx - 1) // 2] = 0
    cases = {"per-edge, random PEC": (eps, kap, True), "per-edge, six materials": (pal_e[mat], pal_k[mat], True), "raw": (eps, kap, False)}
    seen = set()
    for name, (e_r, k_r, prefer) in cases.items():
        engs = []
        for lib, fl in ((hip_lib, flags), (oracle_lib, 0)):
            e = capi.Engine(lib, nx, ny, nz, dt, max_steps=8, flags=fl)
            emet, hmet = eco.pack_metric_tables(*eco.metric_lists(grid, dt), grid, 0, nz)
            e.build_operator(grid.d, e_r, k_r, pec, const.EPS0, eco.lumped_overrides(grid, e_r, k_r, pec, dt, lumped), emet, hmet,
                             prefer_classes=prefer)
            frng = np.random.default_rng(3)
            for kind in (0, 1):
                for c in range(3):
                    e.set_field(kind, c, (1e-3 * frng.standard_normal(e.local_shape)).astype(np.float32))
            e.run(60)
            engs.append(e)
        seen.add(engs[0].operator_form()[0])
        fh, fo = engs[0].fields(), engs[1].fields()
        assert np.isfinite(fo).all() and np.abs(fo).max() > 0
        assert np.array_equal(fh, fo), f"{name} ({engs[0].operator_form()}): {np.abs(fh - fo).max():.3e}"
    assert seen == {"classes", "raw"}, seen

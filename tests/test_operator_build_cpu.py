"""fdtd_build_operator of the oracle (plain C) against the numpy formulation that specifies the operator
(ecoperator.build_operator): every coefficient identical to the bit — uniform and graded meshes, lossy materials,
PEC edges, lumped edges, z-slabs."""
import numpy as np
import pytest

from conftest import pkg
from helpers import patch_sim
from opbuild_cases import random_scene, engine_with_built_operator, same_bits


@pytest.mark.parametrize("seed,shape,graded,nmat", [(1, (21, 18, 15), False, 3), (2, (24, 19, 17), True, 8), (3, (9, 8, 7), True, 5)])
@pytest.mark.parametrize("world,rank", [(1, 0), (3, 1), (3, 2)])
def test_oracle_build_matches_numpy_spec(oracle_lib, seed, shape, graded, nmat, world, rank):
    eco = pkg("ecoperator")
    grid, eps, kap, pec, lumped = random_scene(seed, shape, graded, nmat)
    dt = grid.courant_dt()
    e, k0, nk = engine_with_built_operator(oracle_lib, grid, eps, kap, pec, lumped, dt, rank=rank, world=world)
    spec = eco.build_operator(grid, eps, kap, pec, dt, lumped).raw(k0, nk)
    got = e.get_operator()
    for name, a, b in zip(("vv", "vi", "ii", "iv"), got, spec):
        assert same_bits(a, b), name
    assert np.abs(spec[1]).max() > 0 and (spec[0] != 0).any()
    form, ncls = e.operator_form()
    op = eco.build_operator(grid, eps, kap, pec, dt, lumped)
    pairs = np.unique(np.stack([op.vv[:, k0:k0 + nk].ravel().view(np.uint32), op.m[:, k0:k0 + nk].ravel().view(np.uint32)]), axis=1).shape[1]
    assert (form, ncls) == (("classes", pairs) if pairs <= 256 else ("raw", 0))
    if seed == 2:
        assert form == "raw"       # graded mesh x 8 materials: far beyond 256 classes
    if seed == 1:
        assert form == "classes"


def test_lumped_edges_are_overridden(oracle_lib):
    eco = pkg("ecoperator")
    grid, eps, kap, pec, lumped = random_scene(5, (14, 13, 12), False, 2, n_lumped=4)
    dt = grid.courant_dt()
    e, _, _ = engine_with_built_operator(oracle_lib, grid, eps, kap, pec, lumped, dt)
    e0, _, _ = engine_with_built_operator(oracle_lib, grid, eps, kap, pec, [], dt)
    vv, vv0 = e.get_operator()[0], e0.get_operator()[0]
    diff = np.argwhere(vv != vv0)
    assert {tuple(d) for d in diff} == {(le.comp, le.k, le.j, le.i) for le in lumped}
    assert (vv[vv != vv0] < vv0[vv != vv0]).all()      # a conductance lowers vv


def test_patch_scene_operator_and_fields_same_on_both_set_up_paths(oracle_lib):
    """The product set-up (engine builds the operator) and the host set-up (numpy arrays uploaded) give the same
    operator and, after 150 steps with the port pulse, the same fields."""
    sims = [patch_sim(40, 36, 24, nr_ts=150), patch_sim(40, 36, 24, nr_ts=150)]
    sims[1].device_operator = False
    engs = [s.build(oracle_lib) for s in sims]
    for a, b in zip(engs[0].get_operator(), engs[1].get_operator()):
        assert same_bits(a, b)
    for e in engs:
        e.run(150)
    assert np.array_equal(engs[0].fields(), engs[1].fields()) and np.abs(engs[0].fields()).max() > 0
    assert sims[0].operator_form == sims[1].operator_form == "classes"


def test_build_operator_argument_errors(oracle_lib):
    capi = pkg("_capi")
    grid, eps, kap, pec, lumped = random_scene(7, (10, 9, 8), False, 2)
    dt = grid.courant_dt()
    with pytest.raises(ValueError):
        engine_with_built_operator(oracle_lib, grid, eps[:-1], kap[:-1], pec, lumped, dt)
    e, _, _ = engine_with_built_operator(oracle_lib, grid, eps, kap, pec, [], dt)
    eco, const = pkg("ecoperator"), pkg("constants")
    emet, hmet = eco.pack_metric_tables(*eco.metric_lists(grid, dt), grid)
    bad = (np.array([10 ** 9], np.int64), np.array([0], np.int8), np.array([0.5], np.float32), np.array([1.0], np.float32))
    with pytest.raises(capi.FdtdError, match="out of range"):
        e.build_operator(grid.d, eps, kap, pec, const.EPS0, bad, emet, hmet)

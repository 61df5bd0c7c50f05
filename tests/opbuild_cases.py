"""Scenes for the operator-build parity tests (CPU: oracle C vs the numpy spec; GPU: HIP vs oracle)."""
import numpy as np
from conftest import pkg


def random_scene(seed, shape, graded, nmat, n_lumped=3, pec_frac=0.05):
    """Random mesh / materials / PEC edges / lumped edges.  graded + many materials -> far more than 256 distinct
    coefficient pairs (raw form); uniform mesh + few materials -> class form."""
    rng = np.random.default_rng(seed)
    nx, ny, nz = shape
    grid_m, eco = pkg("grid"), pkg("ecoperator")

    def lines(n):
        d = np.full(n - 1, 1e-3)
        if graded:
            d = d * rng.uniform(0.6, 1.5, n - 1)
        return np.concatenate([[0.0], np.cumsum(d)])

    grid = grid_m.RectGrid(lines(nx), lines(ny), lines(nz))
    eps_pal = np.array([1.0, 4.3, 2.2, 9.8, 3.38, 6.15, 10.2, 1.5])[:nmat]
    kap_pal = np.array([0.0, 2.3e-3, 0.0, 1e-2, 5e-4, 0.0, 3e-3, 0.2])[:nmat]
    mat = rng.integers(0, nmat, size=(nz - 1, ny - 1, nx - 1))
    # blocky materials (boxes), like a voxelised scene, plus some single-cell noise
    blk = rng.integers(0, nmat, size=((nz + 6) // 7, (ny + 6) // 7, (nx + 6) // 7))
    big = np.repeat(np.repeat(np.repeat(blk, 7, 0), 7, 1), 7, 2)[: nz - 1, : ny - 1, : nx - 1]
    mat = np.where(rng.random(mat.shape) < 0.9, big, mat)
    eps, kap = eps_pal[mat].astype(np.float64), kap_pal[mat].astype(np.float64)
    pec = rng.random((3, nz, ny, nx)) < pec_frac
    lumped = []
    for _ in range(n_lumped):
        c = int(rng.integers(0, 3))
        i, j, k = (int(rng.integers(2, n - 2)) for n in (nx, ny, nz))
        pec[c, k, j, i] = False
        lumped.append(eco.LumpedEdge(c, i, j, k, float(rng.uniform(0.005, 0.05))))
    return grid, eps, kap, pec, lumped


def engine_with_built_operator(lib, grid, eps, kap, pec, lumped, dt, *, rank=0, world=1, prefer_classes=True):
    capi, eco, simm, const = pkg("_capi"), pkg("ecoperator"), pkg("simulation"), pkg("constants")
    nx, ny, nz = grid.shape
    k0, nk = simm.slab_range(nz, world, rank)
    e = capi.Engine(lib, nx, ny, nz, dt, k0=k0, nk=nk, rank=rank, world=world, max_steps=8)
    emet, hmet = eco.pack_metric_tables(*eco.metric_lists(grid, dt), grid, k0, nk)
    e.build_operator(grid.d, eps, kap, pec, const.EPS0, eco.lumped_overrides(grid, eps, kap, pec, dt, lumped), emet, hmet,
                     prefer_classes=prefer_classes)
    return e, k0, nk


def same_bits(a, b):
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))

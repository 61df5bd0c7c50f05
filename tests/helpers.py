"""Shared builders for the parity tests: the same seeded/synthetic set-up drives the HIP library
and the oracle through one wrapper (fdtd-solver-antennas_amd/_capi.Engine)."""
import numpy as np
from conftest import pkg


def patch_sim(nx, ny, nz, *, boundary="CPML", cpml_cells=8, nr_ts=400, use_classes=True, nf2ff=True,
              f0=2.45e9, nf2ff_freqs=None, nf2ff_mode="dft"):
    wl, sc, sim = pkg("workloads"), pkg("scene"), pkg("simulation")
    w = wl.patch_workload("test", nx=nx, ny=ny, nz=nz, f0=f0)
    vox = sc.voxelize(w.scene, w.grid)
    return sim.Simulation(w.grid, vox, f0=w.f0, fc=w.fc, boundary=boundary, cpml_cells=cpml_cells,
                          nr_ts=nr_ts, nf2ff_freqs=(nf2ff_freqs if nf2ff_freqs is not None else [w.f0]) if nf2ff else None,
                          use_classes=use_classes, nf2ff_mode=nf2ff_mode)


def seeded_fields(engine, seed=0, scale=1e-3):
    """Fill all six components with seeded noise (exercises every stencil term at once)."""
    rng = np.random.default_rng(seed)
    for kind in (0, 1):
        for c in range(3):
            engine.set_field(kind, c, (scale * rng.standard_normal(engine.local_shape)).astype(np.float32))


def rel_l2(a, b):
    a = np.asarray(a); b = np.asarray(b)
    n = np.linalg.norm(b.ravel())
    return float(np.linalg.norm((a - b).ravel()) / (n if n > 0 else 1.0))

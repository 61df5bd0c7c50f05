"""An EXACT identity of the algorithm the oracle restates (and the HIP kernels reproduce bit for bit) — the discrete Poynting theorem of the
leapfrog EC-FDTD scheme (SURVEY §8c equations: C dV/dt + G V = curl I, L dI/dt = -curl^T V, semi-implicit loss):

    Q_s = 1/2 sum_e C_e (V_e^s)^2  +  1/2 sum_f L_f I_f^s I_f^(s-1)          Q_(s+1) - Q_s = - dt sum_e G_e ((V_e^(s+1) + V_e^s) / 2)^2

for ANY mesh grading, dielectric and conductivity distribution, PEC sheets and lumped resistors, inside PEC walls and without sources — it holds
only if the E update's curl and the H update's curl are exact negative transposes of each other AND every coefficient (vv, vi, iv, the lumped
G folded into an edge) is what the equations say.  C, L, G are read back from the operator the engine built (C = dt (1 + vv) / (2 vi),
L = dt / iv, G dt = 2 C (1 - vv) / (1 + vv)).  Checked on the double build of the oracle to 1e-12 (round-off only) and on the float32 oracle to
1e-5: a KAT with no physics band in it.  Not a pin against openEMS (nothing here can be): a pin of the restatement against its own equations."""
import ctypes

import numpy as np
import pytest

from conftest import pkg
from helpers import load_oracle_f64, build_f64


def _scene(nx=26, ny=22, nz=20):
    """The fixed patch scene (FR-4 substrate with conductivity, PEC patch and ground, a lumped 50 Ohm resistor — the port, NOT excited) on a
    mesh graded in all three directions, PEC walls."""
    wl, sc, simm = pkg("workloads"), pkg("scene"), pkg("simulation")
    grid_mod = pkg("grid")
    w = wl.patch_workload("inv", nx=nx, ny=ny, nz=nz)
    rng = np.random.default_rng(3)

    def graded(lines):        # jitter the interior lines by up to +-30 % of a cell: neighbouring cells differ by up to ~2x
        l = np.array(lines, float)
        d = np.diff(l)
        l[1:-1] += rng.uniform(-0.3, 0.3, l.size - 2) * np.minimum(d[:-1], d[1:])
        return l
    grid = grid_mod.RectGrid(graded(w.grid.x), graded(w.grid.y), w.grid.z)
    port = w.scene.ports[0]
    port.excite = 0.0         # the resistor stays (a lumped G on the port's edges), the source goes
    vox = sc.voxelize(w.scene, grid)
    return simm.Simulation(grid, vox, f0=w.f0, fc=w.fc, boundary="PEC", nr_ts=200, nf2ff_freqs=None)


def _identity(lib, e, sim, steps, dbl):
    n = int(np.prod(e.local_shape))
    f64 = np.float64

    def get_op():
        if dbl:
            out = [np.empty(3 * n, f64) for _ in range(4)]
            assert lib.fdtd_oracle_get_operator_f64(e._ctx, *[a.ctypes.data_as(ctypes.c_void_p) for a in out]) == 0
            return out
        return [np.asarray(a, f64).ravel() for a in e.get_operator()]

    def get_fields(kind):
        if dbl:
            out = np.empty((3, n), f64)
            for c in range(3):
                assert lib.fdtd_oracle_get_field_f64(e._ctx, kind, c, out[c].ctypes.data_as(ctypes.c_void_p)) == 0
            return out.ravel()
        return np.concatenate([np.asarray(e.get_field(kind, c), f64).ravel() for c in range(3)])
    vv, vi, ii, iv = get_op()
    assert np.all(ii == 1.0)
    live_e, live_h = vi != 0, iv != 0
    dt = sim.dt
    C = np.where(live_e, dt * (1 + vv) / (2 * np.where(live_e, vi, 1.0)), 0.0)
    L = np.where(live_h, dt / np.where(live_h, iv, 1.0), 0.0)
    Gdt = np.where(live_e, 2 * C * (1 - vv) / (1 + vv), 0.0)
    assert Gdt.max() > 0 and Gdt.min() >= 0         # the lossy substrate and the resistor are in
    # random initial fields on the live unknowns (dead edges / faces at zero), one timestep to have I^(s-1)
    rng = np.random.default_rng(11)
    for kind, live in ((0, live_e), (1, live_h)):
        for c in range(3):
            g = (rng.standard_normal(n) * live[c * n:(c + 1) * n]).astype(np.float32)
            e.set_field(kind, c, g.reshape(e.local_shape))
    V, I_prev = get_fields(0), get_fields(1)
    e.run(1)
    V, I = get_fields(0), get_fields(1)
    worst, q0 = 0.0, None
    for _ in range(steps):
        q = 0.5 * np.sum(C * V * V) + 0.5 * np.sum(L * I * I_prev)
        q0 = q if q0 is None else q0
        e.run(1)
        V2, I2 = get_fields(0), get_fields(1)
        q2 = 0.5 * np.sum(C * V2 * V2) + 0.5 * np.sum(L * I2 * I)
        loss = np.sum(Gdt * (0.5 * (V + V2)) ** 2)
        worst = max(worst, abs(q2 - q + loss) / q0)
        V, I_prev, I = V2, I, I2
    return worst, q0, q2


def test_discrete_poynting_theorem_double_build(oracle_lib):
    lib64 = load_oracle_f64()
    lib64.fdtd_oracle_get_field_f64.restype = ctypes.c_int
    lib64.fdtd_oracle_get_operator_f64.restype = ctypes.c_int
    sim = _scene()
    e = build_f64(sim, lib64, double_tables=True)
    worst, q0, q_end = _identity(lib64, e, sim, 120, True)
    assert q_end < (1 - 1e-5) * q0       # energy really left through the conductivity and the resistor (4e-5 of it in 120 timesteps) ...
    assert worst < 1e-12, worst          # ... and every timestep's balance closes to round-off


def test_discrete_poynting_theorem_float32_oracle(oracle_lib):
    sim = _scene()
    e = sim.build(oracle_lib)
    worst, q0, q_end = _identity(oracle_lib, e, sim, 120, False)
    assert q_end < (1 - 1e-5) * q0 and worst < 1e-7, worst


def _cavity(n=16, delta=2.0e-3):
    sc, simm, grid_mod = pkg("scene"), pkg("simulation"), pkg("grid")
    lines = np.arange(n + 1) * delta
    grid = grid_mod.RectGrid(lines, lines * 0.75, lines * 1.25)          # cells of 2.0 x 1.5 x 2.5 mm: a different spacing per axis
    vox = sc.voxelize(sc.Scene(unit=1.0), grid)
    return simm.Simulation(grid, vox, f0=1e9, fc=5e8, boundary="PEC", nr_ts=400, nf2ff_freqs=None), grid


@pytest.mark.parametrize("dbl", [True, False])
def test_numerical_dispersion_relation_is_exact(oracle_lib, dbl):
    _dispersion(load_oracle_f64() if dbl else oracle_lib, dbl)


def _dispersion(lib, dbl, flags=0):
    """"numerical dispersion vs analytic".  In a PEC box on a uniform grid E_y = sin(k_x x_i) sin(k_z z_k) is an EXACT eigenvector
    of the discrete curl-curl operator, so every edge voltage obeys V(n+1) + V(n-1) = 2 cos(w dt) V(n) with the Yee dispersion relation
    sin^2(w dt / 2) = (c dt)^2 [sin^2(k_x dx / 2) / dx^2 + sin^2(k_z dz / 2) / dz^2] — which sees what the energy identity cannot: the constants
    (eps0, mu0, dt) and the metric in the coefficients.  Mode (3, 0, 2) on 16 cells per side (5 - 8 cells per half wavelength): the discrete
    frequency lies 1-2 % below the continuum's, and the engine follows the DISCRETE one to round-off."""
    consts = pkg("constants")
    sim, grid = _cavity()
    e = build_f64(sim, lib, double_tables=True) if dbl else sim.build(lib, flags=flags)
    nx, ny, nz = grid.shape
    m, p = 3, 2
    dx, dz = grid.x[1] - grid.x[0], grid.z[1] - grid.z[0]
    kx, kz = m * np.pi / (grid.x[-1] - grid.x[0]), p * np.pi / (grid.z[-1] - grid.z[0])
    shape = np.sin(kx * (grid.x - grid.x[0]))[None, None, :] * np.sin(kz * (grid.z - grid.z[0]))[:, None, None] * np.ones((1, ny, 1))
    shape[:, -1, :] = 0.0                                   # (the y edge from the last node does not exist)
    if dbl:      # (the ABI's setter rounds to float32: the mode would carry 6e-8 of every other mode)
        lib.fdtd_oracle_set_field_f64.restype = ctypes.c_int
        sh64 = np.ascontiguousarray(shape, np.float64)
        assert lib.fdtd_oracle_set_field_f64(e._ctx, 0, 1, sh64.ctypes.data_as(ctypes.c_void_p)) == 0
    else:
        e.set_field(0, 1, shape.astype(np.float32))
    s2 = (consts.C0 * sim.dt) ** 2 * (np.sin(kx * dx / 2) ** 2 / dx ** 2 + np.sin(kz * dz / 2) ** 2 / dz ** 2)
    w_num = 2.0 * np.arcsin(np.sqrt(s2)) / sim.dt
    w_cont = consts.C0 * np.hypot(kx, kz)
    assert 0.005 < 1.0 - w_num / w_cont < 0.03               # a coarse mode: the grid's own frequency, visibly below the continuum's

    def vy():
        if dbl:
            out = np.empty(nx * ny * nz)
            lib.fdtd_oracle_get_field_f64.restype = ctypes.c_int
            assert lib.fdtd_oracle_get_field_f64(e._ctx, 0, 1, out.ctypes.data_as(ctypes.c_void_p)) == 0
            return out.reshape(nz, ny, nx)
        return np.asarray(e.get_field(0, 1), np.float64)
    hist = [vy()]
    for _ in range(200):
        e.run(1)
        hist.append(vy())
    a = np.abs(hist[0]).max()
    resid = max(np.abs(hist[n + 1] + hist[n - 1] - 2.0 * np.cos(w_num * sim.dt) * hist[n]).max() for n in range(1, 200)) / a
    assert resid < (1e-13 if dbl else 2e-6), resid
    # the other components stay zero: the mode is an eigenvector, nothing leaks
    assert max(np.abs(np.asarray(e.get_field(0, c))).max() for c in (0, 2)) <= (1e-12 if dbl else 1e-5) * a
    # and with the CONTINUUM frequency the same recurrence is off by the dispersion error, orders of magnitude above round-off
    off = max(np.abs(hist[n + 1] + hist[n - 1] - 2.0 * np.cos(w_cont * sim.dt) * hist[n]).max() for n in range(1, 200)) / a
    assert off > 1e-3

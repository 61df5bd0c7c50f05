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

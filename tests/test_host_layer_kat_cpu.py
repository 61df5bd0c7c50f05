"""Known-answer tests for the HOST layer both engines share.

Every `-m gpu` parity test compares libfdtd_hip.so with the oracle THROUGH the same Python set-up code, so an error in
that code is invisible to them.  openEMS itself is unobtainable (parity unpinned); these tests tie the shared
layer to closed-form answers instead (DESIGN.md §2 lists which module each one guards):

  * a 50-ohm microstrip line drawn through openems_api, two lumped ports  ->  Z0 and eps_eff from the ABCD matrix of the
    line against Wheeler / Hammerstad (the reference's own formulas: microstrip.py:84-112, physics.py:19-28)
    [openems_api scene calls, mesher thirds rule + smoothing, scene voxeliser, ecoperator material averaging and
     lumped-R folding, excitation, port probes, LumpedPort.CalcPort]
  * CPML terminating a DIELECTRIC-filled parallel-plate line (and one half filled, the interface running into the
    layers, as the substrate of the legacy scene does: solver_fdtd_openems.py:188-203)  ->  reflection below -60 dB
    [cpml.build_cpml profiles — sigma_opt is taken with the free-space impedance everywhere]
  * LumpedPort.CalcPort on synthetic U/I series of a line terminated by a series RLC  ->  S11(f) against the closed form
    [CalcPort: DFT scaling, the half-step offset of the current samples, incident / reflected split]
"""
import types

import numpy as np
import pytest

from conftest import pkg

C0 = 299792458.0


# ---------------------------------------------------------------------------------------------------------------------
# (a) microstrip line: Z0 and eps_eff
# ---------------------------------------------------------------------------------------------------------------------
def _microstrip_line(lib, sim_path, length=30.0, eps_r=4.3, h=1.6, nr_ts=4000):
    oa, pd = pkg("openems_api"), pkg("patch_design")
    f0, fc = 1.5e9, 1.4e9
    W = pd.calculate_microstrip_width(f0, eps_r, h * 1e-3) * 1e3          # 3.114 mm for eps_r 4.3, h 1.6
    fdtd = oa.openEMS(NrTS=nr_ts, EndCriteria=0, lib=lib)
    fdtd.SetGaussExcite(f0, fc)
    fdtd.SetBoundaryCond(["MUR"] * 6)
    csx = oa.ContinuousStructure()
    fdtd.SetCSX(csx)
    mesh = csx.GetGrid()
    mesh.SetDeltaUnit(1e-3)
    res = C0 / (f0 + fc) / 1e-3 / 40.0
    sx, sy = length + 40.0, 40.0
    mesh.AddLine("x", [-sx / 2 - 15, sx / 2 + 15])
    mesh.AddLine("y", [-sy / 2 - 15, sy / 2 + 15])
    mesh.AddLine("z", [-12.0, 25.0])
    sub = csx.AddMaterial("substrate", epsilon=eps_r, kappa=0.0)
    sub.AddBox(priority=0, start=[-sx / 2, -sy / 2, 0.0], stop=[sx / 2, sy / 2, h])
    mesh.AddLine("z", np.linspace(0.0, h, 5).tolist())
    gnd = csx.AddMetal("gnd")
    gnd.AddBox([-sx / 2, -sy / 2, 0.0], [sx / 2, sy / 2, 0.0], priority=10)
    fdtd.AddEdges2Grid(dirs="xy", properties=gnd)
    strip = csx.AddMetal("strip")
    strip.AddBox([-length / 2, -W / 2, h], [length / 2, W / 2, h], priority=10)
    fdtd.AddEdges2Grid(dirs="y", properties=strip, metal_edge_res=W / 4.0)   # thirds rule across the strip
    mesh.AddLine("x", [-length / 2, length / 2])
    p1 = fdtd.AddLumpedPort(1, 50.0, [-length / 2, 0.0, 0.0], [-length / 2, 0.0, h], "z", 1.0, priority=5, edges2grid="xy")
    p2 = fdtd.AddLumpedPort(2, 50.0, [length / 2, 0.0, 0.0], [length / 2, 0.0, h], "z", 0.0, priority=5, edges2grid="xy")
    mesh.SmoothMeshLines("all", res, 1.4)
    fdtd.Run(sim_path, verbose=0, cleanup=True)
    return fdtd, p1, p2, W


def _abcd_of_line(p1, p2, f):
    """Z0 and beta*l of the line between two lumped ports from ONE excitation: the two-port is symmetric and reciprocal,
    U1 = A U2 + B Io, I1 = C U2 + A Io with A^2 - BC = 1, Io = current leaving the line into the load = -I2 (the current
    probe of a port counts ground -> strip).  A = cos(beta l), B = j Z0 sin(beta l), C = j sin(beta l) / Z0."""
    p1.CalcPort("", f)
    p2.CalcPort("", f)
    U1, I1, U2, Io = p1.uf_tot, p1.if_tot, p2.uf_tot, -p2.if_tot
    A = (U1 * I1 + U2 * Io) / (U1 * Io + U2 * I1)
    B = (U1 - A * U2) / Io
    Cc = (I1 - A * Io) / U2
    z0 = np.sqrt(B / Cc)
    bl = np.unwrap(np.arctan2((B / (1j * z0)).real, A.real))
    return z0, bl, A


def test_microstrip_line_z0_and_eps_eff(oracle_lib, tmp_path):
    """Measured on the oracle: Z0 = 50.3 .. 48.5 ohm over 1-2 GHz (Wheeler's width for 50 ohm), eps_eff = 3.27 .. 3.30
    against Hammerstad's 3.266.  eps_eff comes from the phase DIFFERENCE of a 30 mm and a 50 mm line, which cancels the
    open ends and the vertical ports (taken from one line alone they add 2.5 mm of electrical length: 3.83)."""
    pd = pkg("patch_design")
    eps_r, h = 4.3, 1.6
    f = np.linspace(1.0e9, 2.0e9, 11)
    res = {}
    for length in (30.0, 50.0):
        fdtd, p1, p2, W = _microstrip_line(oracle_lib, str(tmp_path / f"msl{int(length)}"), length, eps_r, h, nr_ts=5000)
        res[length] = _abcd_of_line(p1, p2, f)
        # the passive port is its resistor: U = -R I in the port's own current orientation
        zl = p2.uf_tot / p2.if_tot
        assert np.all(np.abs(zl.real + 50.0) < 0.5) and np.all(np.abs(zl.imag) < 2.5), zl
        u = p1.u_data.ui_val[0]
        assert np.abs(u[-200:]).max() < 1e-4 * np.abs(u).max()             # the pulse has left through the two loads
    assert abs(W - 3.1144) < 1e-3
    z0, bl, A = res[30.0]                                                  # beta*l = 1.1 .. 2.3 rad: far from 0 and pi
    assert np.max(np.abs(A.imag)) < 0.02                                   # lossless line (kappa = 0, PEC strip)
    assert np.all(np.abs(z0.real - 50.0) < 0.05 * 50.0) and np.all(np.abs(z0.imag) < 1.0), z0
    beta = (res[50.0][1] - res[30.0][1]) / 20e-3
    eps_eff = (beta * C0 / (2 * np.pi * f)) ** 2
    ee_ref = pd.effective_eps(eps_r, h * 1e-3, W * 1e-3)                   # Hammerstad, physics.py:19-28
    assert np.all(np.abs(eps_eff / ee_ref - 1.0) < 0.03), (eps_eff, ee_ref)
    # and a 50-ohm line between 50-ohm ports reflects little
    p1 = fdtd._ports[0]
    assert np.max(20 * np.log10(np.abs(p1.uf_ref / p1.uf_inc))) < -20.0


# ---------------------------------------------------------------------------------------------------------------------
# (b) CPML terminating a dielectric
# ---------------------------------------------------------------------------------------------------------------------
def _plate_line(lib, nx, nz, boundary, steps, src, prb, eps_fill, half=False, cells=10, d=2e-3):
    """Parallel-plate line: PEC plates normal to y, two cells apart — the TEM family (Ey, Hx, Hz), no variation along
    y: a 2-D problem in the x-z plane.  Filled with eps_fill (half: only below the source plane)."""
    g = pkg("grid").RectGrid(np.arange(nx) * d, np.arange(3) * d, np.arange(nz) * d)
    eps = np.full((nz - 1, 2, nx - 1), float(eps_fill))
    if half:
        eps[src[1]:, :, :] = 1.0
    kap = np.zeros_like(eps)
    pec = np.zeros((3, nz, 3, nx), bool)
    bc = [boundary, boundary, "PEC", "PEC", boundary, boundary]
    s = pkg("simulation").Simulation(g, pkg("scene").VoxelScene(eps, kap, pec, []), f0=3e9, fc=2.5e9, boundary=bc,
                                     cpml_cells=cells, nr_ts=steps)
    e = s.build(lib)
    e.add_source([g.flat(src[0], 1, src[1]), g.flat(src[0], 0, src[1])], [1, 1], [1.0, 1.0])
    pid = e.add_probe(0, [g.flat(prb[0], 1, prb[1])], [1], [1.0])
    e.run(steps)
    return e.get_probe(pid)


@pytest.mark.parametrize("fill,half", [(4.3, False), (4.3, True)])
def test_cpml_terminating_a_dielectric(oracle_lib, fill, half):
    """Reflection = difference to a run in a PEC box whose walls cannot echo within the window.  Wave speed in the fill:
    0.577 / sqrt(4.3) = 0.28 cells per step (0.577 in the air half); 700 steps."""
    n, steps, off = 64, 700, 230
    small = _plate_line(oracle_lib, n, n, "CPML", steps, (32, 32), (40, 27), fill, half)
    big = _plate_line(oracle_lib, n + 2 * off, n + 2 * off, "PEC", steps, (32 + off, 32 + off), (40 + off, 27 + off), fill, half)
    err = np.max(np.abs(small - big)) / np.max(np.abs(big))
    # measured on the oracle: eps_r 1: -91.0 dB, 4.3: -91.6 dB, 4.3 half filled: -100.8 dB, 9: -87.6 dB — the free-space
    # sigma_opt of cpml.py needs no sqrt(eps_r) scaling at these layer thicknesses (the bar VERDICT r2 asked for: -40 dB)
    assert 20 * np.log10(err) < -60.0, f"CPML in front of eps_r {fill} ({'half' if half else 'full'}): reflection {20 * np.log10(err):.1f} dB"


# ---------------------------------------------------------------------------------------------------------------------
# (c) CalcPort against a closed-form S11
# ---------------------------------------------------------------------------------------------------------------------
def test_calcport_on_a_series_rlc_load():
    """u(t), i(t) of a 50-ohm source line terminated by a series RLC, synthesised from the closed-form reflection
    coefficient; CalcPort must give that coefficient back (the current is sampled half a step after the voltage)."""
    oa = pkg("openems_api")
    Z0, R, L, Cap = 50.0, 20.0, 12e-9, 0.35e-12              # resonance 2.456 GHz, |S11| min = (50-20)/(50+20)
    dt, n = 2.0e-12, 1 << 16
    t = np.arange(n) * dt
    f0, fc = 2.45e9, 1.2e9
    t0 = 9.0 / (2 * np.pi * fc)
    a = np.cos(2 * np.pi * f0 * (t - t0)) * np.exp(-(2 * np.pi * fc * t / 3.0 - 3.0) ** 2)    # incident wave (volts)
    fr = np.fft.rfftfreq(n, dt)
    w = 2 * np.pi * np.maximum(fr, 1.0)
    ZL = R + 1j * w * L + 1.0 / (1j * w * Cap)
    gam = (ZL - Z0) / (ZL + Z0)
    Af = np.fft.rfft(a)
    u = np.fft.irfft(Af * (1 + gam), n)
    i_half = np.fft.irfft(Af * (1 - gam) / Z0 * np.exp(1j * 2 * np.pi * fr * 0.5 * dt), n)      # sampled at (k + 1/2) dt
    assert np.abs(u[-2000:]).max() < 1e-6 * np.abs(u).max()     # the ringing has died inside the window: no wrap-around
    fdtd = oa.openEMS()
    fdtd.sim = types.SimpleNamespace(dt=dt)
    port = oa.LumpedPort(fdtd, 1, Z0, [0, 0, 0], [0, 0, 1], 2, 1.0)
    fdtd._ports, fdtd._u_i = [port], [(u, i_half)]
    f = np.linspace(1.5e9, 3.5e9, 81)
    port.CalcPort("", f)
    s11 = port.uf_ref / port.uf_inc
    wz = 2 * np.pi * f
    zl = R + 1j * wz * L + 1.0 / (1j * wz * Cap)
    exact = (zl - Z0) / (zl + Z0)
    assert np.max(np.abs(s11 - exact)) < 1e-6
    k = int(np.argmin(np.abs(s11)))
    assert abs(f[k] - 1.0 / (2 * np.pi * np.sqrt(L * Cap))) < (f[1] - f[0]) and abs(abs(s11[k]) - 30.0 / 70.0) < 1e-3
    # incident / accepted power bookkeeping
    assert np.allclose(port.P_acc, port.P_inc - port.P_ref, rtol=1e-9)
    assert np.allclose(port.uf_inc, 2.0 * dt * np.exp(-2j * np.pi * np.outer(f, t)) @ a, rtol=1e-6, atol=1e-12 * np.abs(port.uf_inc).max())

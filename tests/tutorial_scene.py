"""The one scene of the reference with a community-known answer: its `test_openems.py:19-40,37-99` builds the openEMS
"Simple Patch Antenna" tutorial (32 x 40 mm patch, eps_r 3.38, h 1.524 mm, 60 x 60 mm substrate, f0 2 GHz, fc 1 GHz,
PML on all faces `[3]*6`, NrTS 60000, EndCriteria 1e-5, 50-ohm lumped port at x = -6 mm) and only checks that Run()
returns.  Here the same calls go through this package's mirror of that API (openems_api), so the scene can be stepped
on the HIP library or on the oracle and its S11 dip and directivity compared with the physics band SURVEY §8(c) names.

Not a copy of the reference script: the numbers are its inputs (data), the calls are the public openEMS API."""
import numpy as np

from conftest import pkg


def build_and_run(lib, sim_path, nr_ts=60000, end_criteria=1e-5, verbose=0, f_ff=None, post=True, loss_tangent=1e-3, full_sphere=False, **fdtd_kw):
    """f_ff: evaluate the far field THERE instead of at the S11 dip; post=False: Run() only (a rank that leaves the
    post-processing to rank 0); fdtd_kw: backend options of openems_api.openEMS (rank=, world=, comm=, ...)."""
    oa = pkg("openems_api")
    C0, EPS0 = oa.physical_constants.C0, oa.physical_constants.EPS0
    patch_w, patch_l = 32.0, 40.0                 # mm: x (resonant), y
    eps_r, h, sub = 3.38, 1.524, 60.0
    feed_x, feed_R = -6.0, 50.0
    box = np.array([200.0, 200.0, 150.0])
    f0, fc = 2e9, 1e9
    fdtd = oa.openEMS(NrTS=nr_ts, EndCriteria=end_criteria, lib=lib, **fdtd_kw)
    fdtd.SetGaussExcite(f0, fc)
    fdtd.SetBoundaryCond([3] * 6)
    csx = oa.ContinuousStructure()
    fdtd.SetCSX(csx)
    mesh = csx.GetGrid()
    mesh.SetDeltaUnit(1e-3)
    res = C0 / (f0 + fc) / 1e-3 / 20.0
    mesh.AddLine("x", [-box[0] / 2, box[0] / 2])
    mesh.AddLine("y", [-box[1] / 2, box[1] / 2])
    mesh.AddLine("z", [-box[2] / 3, box[2] * 2 / 3])
    patch = csx.AddMetal("patch")
    patch.AddBox(priority=10, start=[-patch_w / 2, -patch_l / 2, h], stop=[patch_w / 2, patch_l / 2, h])
    fdtd.AddEdges2Grid(dirs="xy", properties=patch, metal_edge_res=res / 2)
    substrate = csx.AddMaterial("substrate", epsilon=eps_r, kappa=2 * np.pi * f0 * EPS0 * eps_r * loss_tangent)
    substrate.AddBox(priority=0, start=[-sub / 2, -sub / 2, 0.0], stop=[sub / 2, sub / 2, h])
    mesh.AddLine("z", np.linspace(0.0, h, 5).tolist())
    gnd = csx.AddMetal("gnd")
    gnd.AddBox([-sub / 2, -sub / 2, 0.0], [sub / 2, sub / 2, 0.0], priority=10)
    fdtd.AddEdges2Grid(dirs="xy", properties=gnd)
    mesh.AddLine("x", [feed_x])
    mesh.AddLine("z", [0.0, h])
    port = fdtd.AddLumpedPort(1, feed_R, [feed_x, 0.0, 0.0], [feed_x, 0.0, h], "z", 1.0, priority=5, edges2grid="xy")
    mesh.SmoothMeshLines("all", res, 1.4)
    nf2ff = fdtd.CreateNF2FFBox()
    fdtd.Run(sim_path, verbose=verbose, cleanup=True)
    if not post:
        return {"steps": fdtd.stats.steps}
    # post-processing of the tutorial: S11 over f0 +- fc, far field at the dip
    f = np.linspace(max(1e9, f0 - fc), f0 + fc, 401)
    port.CalcPort(sim_path, f)
    s11 = port.uf_ref / port.uf_inc
    s11_dB = 20 * np.log10(np.abs(s11))
    k = int(np.argmin(s11_dB))
    theta = np.arange(0.0, 181.0, 2.0)
    phi = np.arange(0.0, 360.0, 6.0) if full_sphere else [0.0, 90.0]
    ff = nf2ff.CalcNF2FF(sim_path, f[k] if f_ff is None else f_ff, theta, phi, center=[0.0, 0.0, 1e-3])
    return {"f": f, "s11": s11, "s11_dB": s11_dB, "f_dip": float(f[k]), "dip_dB": float(s11_dB[k]),
            "Dmax": float(np.asarray(ff.Dmax)[0]), "E_norm": np.asarray(ff.E_norm[0]), "theta": theta,
            "grid": fdtd.sim.grid.shape, "steps": fdtd.stats.steps, "energy_db": float(fdtd.stats.energy_db),
            "u": np.asarray(port.u_data.ui_val[0]), "i": np.asarray(port.i_data.ui_val[0]),
            "nf2ff_mode": fdtd.sim.nf2ff_mode,
            # absolute powers at the far-field frequency: radiated (flux through the NF2FF box) and accepted at the port
            "Prad": float(np.asarray(ff.Prad)[0]), "P_acc": float(port.P_acc[k]), "P_inc": float(port.P_inc[k])}

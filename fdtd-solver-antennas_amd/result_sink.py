"""Headless result sink: the E/H-plane cuts and summary the reference's GUI derives from a solver
result (gui_app.py:3024-3066: phi = 0 and phi = pi/2 columns of the (ntheta, nphi) dBi grid feed
``PlotFrame.update_2d_patterns(theta, cuts)``; the full grid feeds ``update_3d_pattern``), written as
JSON (and optionally a PNG) instead of Tk canvases.  SURVEY §8(f)-2.
"""
from __future__ import annotations

import json
from typing import Optional

import numpy as np


def principal_cuts(theta, phi, intensity):
    """(theta, E-plane cut at phi~0, H-plane cut at phi~pi/2) exactly as the GUI picks them."""
    th = np.asarray(theta, float)
    ph = np.asarray(phi, float)
    I = np.asarray(intensity, float)
    if I.ndim != 2 or I.shape != (th.size, ph.size):
        raise ValueError("intensity must be (ntheta, nphi)")
    i0 = int(np.argmin(np.abs(ph - 0.0)))
    i90 = int(np.argmin(np.abs(ph - np.pi / 2)))
    return th, I[:, i0], I[:, i90]


def summarize(result) -> dict:
    """JSON-able summary of an FDTDResult / OpenEMSResult-shaped object."""
    th, e_cut, h_cut = principal_cuts(result.theta, result.phi, result.intensity)
    k = np.unravel_index(int(np.argmax(result.intensity)), np.asarray(result.intensity).shape)
    out = {"ok": bool(result.ok), "message": result.message, "is_dBi": bool(result.is_dBi),
           "theta_deg": np.rad2deg(th).tolist(), "e_plane_dBi": e_cut.tolist(), "h_plane_dBi": h_cut.tolist(),
           "max_dBi": float(np.max(result.intensity)),
           "max_dir_deg": [float(np.rad2deg(th[k[0]])), float(np.rad2deg(np.asarray(result.phi)[k[1]]))]}
    for name in ("f_res", "Dmax", "dt"):
        v = getattr(result, name, None)
        if v is not None:
            out[name] = float(v)
    if getattr(result, "s11_dB", None) is not None:
        out["freq_hz"] = np.asarray(result.freq).tolist()
        out["s11_dB"] = np.asarray(result.s11_dB).tolist()
    if getattr(result, "stats", None):
        out["stats"] = result.stats
    return out


def write_result(result, json_path: str, png_path: Optional[str] = None) -> dict:
    data = summarize(result)
    with open(json_path, "w") as fh:
        json.dump(data, fh)
    if png_path:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        fig = plt.figure(figsize=(9, 4))
        ax = fig.add_subplot(1, 2, 1, projection="polar")
        th = np.deg2rad(data["theta_deg"])
        ax.plot(th, data["e_plane_dBi"], label="E-plane (phi=0)")
        ax.plot(-th, data["h_plane_dBi"], label="H-plane (phi=90)")
        ax.set_theta_zero_location("N")
        ax.legend(loc="lower center", fontsize=7)
        if "s11_dB" in data:
            ax2 = fig.add_subplot(1, 2, 2)
            ax2.plot(np.asarray(data["freq_hz"]) / 1e9, data["s11_dB"])
            ax2.set_xlabel("GHz"); ax2.set_ylabel("S11 [dB]"); ax2.grid(True)
        fig.tight_layout()
        fig.savefig(png_path, dpi=110)
        plt.close(fig)
    return data

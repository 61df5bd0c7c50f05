"""Gaussian-modulated excitation pulse.

Restates what ``FDTD.SetGaussExcite(f0, fc)`` (antenna_sim/solver_fdtd_openems_fixed.py:172,
fc = f0/2 at :168) makes the external engine generate ([EXT] openEMS
Excitation::CalcGaussianPulsExcitation):

    s[n] = cos(2 pi f0 (n dt - t0)) * exp(-(2 pi fc n dt / 3 - 3)^2),   t0 = 9/(2 pi fc),
    length = 2 t0 / dt samples.
"""
from __future__ import annotations

import numpy as np


def gauss_pulse_length(fc: float, dt: float) -> int:
    return int(np.ceil(2.0 * 9.0 / (2.0 * np.pi * fc) / dt))


def gauss_pulse(f0: float, fc: float, dt: float, n: int | None = None, dtype=np.float32) -> np.ndarray:
    n = gauss_pulse_length(fc, dt) if n is None else int(n)
    t = np.arange(n, dtype=np.float64) * dt
    t0 = 9.0 / (2.0 * np.pi * fc)
    s = np.cos(2.0 * np.pi * f0 * (t - t0)) * np.exp(-(2.0 * np.pi * fc * t / 3.0 - 3.0) ** 2)
    return s.astype(dtype)


def dft_twiddles(freqs, dt: float, every: int, nsamples: int, offset_steps: float = 0.0) -> np.ndarray:
    """exp(-j 2 pi f t_s) for sample s taken at step every*s (+offset_steps: 0.5 for the H
    half-step), shape [nsamples][nfreq][2] float64 — the table fdtd_set_dft() takes."""
    f = np.atleast_1d(np.asarray(freqs, dtype=np.float64))
    t = (np.arange(nsamples, dtype=np.float64) * every + offset_steps) * dt
    ph = -2.0 * np.pi * np.outer(t, f)
    return np.stack([np.cos(ph), np.sin(ph)], axis=-1)

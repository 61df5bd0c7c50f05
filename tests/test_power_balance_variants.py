"""Power balance of every plugin variant, loss-free: what the port delivers at the resonance (0.5 Re U I*, LumpedPort.CalcPort) against
what leaves through the NF2FF box (nf2ff.Prad) — the one check that sees the ABSOLUTE scale of the two post-processing chains
against each other, on the reference's own scenes (prepare_*: solver_fdtd_openems_fixed.py:113-254 and siblings).

Why not at the reference's defaults: its EndCriteria of 1e-4 (-40 dB, solver_fdtd_openems_fixed.py:171) truncates the port and NF2FF
series of these resonant patches while they still ring; the spectra of the truncated series carry a ripple that reads 88.7 % at the
S11 minimum of the default scene (73 % / 110 % five per cent below / above it) whatever the substrate loss and whatever the
boundary (MUR or CPML).  Run to -60 dB the same scene gives 99.2 % (profiles/r04/power_balance_fixed_scene.txt).  So the bar is taken
on runs to -60 dB with loss_tangent = 0: 0.97 ... 1.03."""
import os

import numpy as np
import pytest

from conftest import pkg


def _balance(prep, res):
    k = int(np.argmin(res.s11_dB))
    fr = float(res.freq[k])
    nfr = prep.nf.CalcNF2FF(prep.sim_path, [fr], np.arange(0.0, 181.0, 6.0), np.arange(0.0, 360.0, 12.0), center=[0, 0, 0])
    acc = sum(float(p.CalcPort(prep.sim_path, np.array([fr])).P_acc[0]) for p in (prep.ports or [prep.port]))
    return float(np.asarray(nfr.Prad)[0]) / acc, fr, float(res.s11_dB[k])


def _variants(s, P, tmp):
    p245 = P.from_user_units(frequency_ghz=2.45, er=4.3, h_mm=1.6, loss_tangent=0.0)
    return {
        "fixed": lambda lib: s.prepare_hip_patch_fixed(p245, work_dir=os.path.join(tmp, "a"), lib=lib),
        "legacy": lambda lib: s.prepare_hip_patch(p245, work_dir=os.path.join(tmp, "f"), lib=lib),
        "microstrip": lambda lib: s.prepare_hip_microstrip_patch(p245, work_dir=os.path.join(tmp, "b"), lib=lib),
        "microstrip_3d": lambda lib: s.prepare_hip_microstrip_patch_3d(p245, work_dir=os.path.join(tmp, "d"), lib=lib),
        # (two elements: the reference's 2 x 2 array at this pitch rings on for more than 150 000 timesteps — -50.7 dB there, 77 % read from
        #  the truncated series — which is a property of the scene, not of the chain)
        "multi_3d": lambda lib: s.prepare_hip_microstrip_multi_3d(
            [s.PatchInstance(f"P{n}", p245, (ix - 0.5) * 0.09, 0.0, 0.0, s.FeedDirection.NEG_X) for n, ix in enumerate([0, 1])],
            work_dir=os.path.join(tmp, "e"), lib=lib),
    }


def _run(name, lib, tmp):
    s, P = pkg("solver_fdtd_hip"), pkg("params").PatchAntennaParams
    prep = _variants(s, P, tmp)[name](lib)
    assert prep.ok, prep.message
    prep.FDTD.EndCriteria = 1e-6
    prep.FDTD.NrTS = max(int(prep.FDTD.NrTS), 200000)     # (the loss-free MUR scenes take 70 000 timesteps to -60 dB)
    res = s.run_prepared_hip(prep, frequency_hz=2.45e9, verbose=0)
    assert res.ok, res.message
    assert res.stats["energy_db"] < -60.0, res.stats
    return _balance(prep, res)


@pytest.mark.parametrize("name", ["fixed", "legacy"])
def test_loss_free_power_balance_cpu(oracle_lib, tmp_path, name):
    eff, fr, dip = _run(name, oracle_lib, str(tmp_path))
    assert dip < -3.0 and 0.97 <= eff <= 1.03, (name, eff, fr, dip)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["fixed", "legacy", "microstrip", "microstrip_3d"])
def test_loss_free_power_balance_every_variant_gpu(hip_lib, tmp_path, name):
    eff, fr, dip = _run(name, hip_lib, str(tmp_path))
    assert 0.97 <= eff <= 1.03, (name, eff, fr, dip)      # (loss-free the microstrip variants are barely matched: S11 minima of -1.2 dB)


@pytest.mark.gpu
def test_multi_3d_port_box_dissipates(hip_lib, tmp_path):
    """The multi-patch variant is NOT loss-free with a loss-free substrate.  Its lumped-port box reaches `ext` = max(0.1, res / 4) mm beyond the
    ground sheet and beyond the patch (solver_fdtd_openems_microstrip_multi_3d.py:498-512, mirrored call for call: z = -1.82 ... +1.82 mm around
    a 1.6 mm substrate), so resistive port edges sit in the open space below / above the conductors and absorb.  Measured on one element, run
    to -60 dB (scratch script of round 4): P_rad / P_acc = 0.659 at the resonance and 0.054 at 2.45 GHz as drawn (MUR and PML_8 alike: 0.659 /
    0.663; two elements 0.605); with the box clipped to the gap (-0.8 ... +0.8 mm) 0.944 and 0.703 — the overhang is most of it, the rest is a
    2.9 x 2.9 mm port whose voltage is sampled on its centre line and whose current around the whole box (not pursued: the reference's geometry).
    Recorded, not barred: below one, and the same for both absorbers."""
    eff, fr, dip = _run("multi_3d", hip_lib, str(tmp_path))
    assert 0.3 < eff < 1.0, (eff, fr, dip)

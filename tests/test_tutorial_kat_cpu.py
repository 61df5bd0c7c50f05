"""Known-answer test on the one scene of the reference with a community-known result: the openEMS "Simple Patch
Antenna" tutorial that /root/reference/test_openems.py:19-99 builds (and only checks for "Run() did not throw").
Built here through openems_api (tests/tutorial_scene.py) and stepped on the oracle — the closest available anchor of
the oracle to openEMS behaviour (S11 dip 2.35-2.60 GHz, D 6-8 dBi; the reference's own Hammerstad
formulas, physics.py:19-38, give 2.51 GHz for the 32 mm resonant length).  Also pins the time-domain NF2FF recorder
against the running DFT on the CPU."""
import numpy as np

from conftest import pkg
from helpers import patch_sim, rel_l2
import tutorial_scene


def test_tutorial_patch_s11_dip_and_directivity(oracle_lib, tmp_path):
    r = tutorial_scene.build_and_run(oracle_lib, str(tmp_path / "tut"))
    # graded mesh of the scene as the reference draws it: ~0.1 Mcell, stops on the -50 dB energy criterion
    assert 40 <= min(r["grid"]) and max(r["grid"]) <= 80
    assert r["steps"] < 60000 and r["energy_db"] < -50.0
    # the dip: a clear resonance inside the band SURVEY §8(c) names for this patch
    assert r["dip_dB"] < -10.0
    assert 2.35e9 <= r["f_dip"] <= 2.60e9, r["f_dip"]
    # directivity at the resonance (possible after the run because the NF2FF faces were recorded in the time domain)
    assert r["nf2ff_mode"] == "record"
    d_dbi = 10.0 * np.log10(r["Dmax"])
    assert 6.0 <= d_dbi <= 8.0, d_dbi
    # broadside pattern: maximum within 10 degrees of theta = 0 in both cuts, back lobe at least 10 dB down
    e = r["E_norm"]
    for col in range(2):
        assert r["theta"][int(np.argmax(e[:, col]))] <= 10.0
        assert 20 * np.log10(e[-1, col] / e[:, col].max()) < -10.0
    # passive one-port: |S11| <= 1 over the whole band (to numerical accuracy)
    assert np.max(np.abs(r["s11"])) < 1.0 + 1e-3


def test_power_balance_of_the_tutorial_patch(oracle_lib, tmp_path):
    """The one check that sees the ABSOLUTE scale of the two post-processing chains against each other (S11 and directivity are ratios
    inside one chain each): the power the port delivers at the resonance, 0.5 Re(U I*) from LumpedPort.CalcPort, against the power that
    leaves through the NF2FF box, nf2ff.Prad — what the openEMS tutorial prints as its radiation efficiency.  Loss-free substrate, PEC
    metal: everything accepted is radiated, 100 % (found at 25 % in round 3: single-sided port spectra, two-sided NF2FF spectra); with
    the tutorial's loss tangent of 1e-3 the dielectric takes Q_rad / Q_d ~ 4 %."""
    free = tutorial_scene.build_and_run(oracle_lib, str(tmp_path / "free"), loss_tangent=0.0)
    eff = free["Prad"] / free["P_acc"]
    assert free["dip_dB"] < -10.0 and 0.97 <= eff <= 1.03, eff
    lossy = tutorial_scene.build_and_run(oracle_lib, str(tmp_path / "lossy"))
    eff_l = lossy["Prad"] / lossy["P_acc"]
    assert 0.90 <= eff_l <= 0.985 and eff_l < eff, (eff_l, eff)
    assert lossy["P_acc"] <= lossy["P_inc"] * (1 + 1e-9)


def test_recorder_equals_running_dft_on_the_oracle(oracle_lib):
    f0 = 2.45e9
    freqs = np.array([0.8 * f0, f0, 1.17 * f0])
    sd = patch_sim(36, 34, 30, nr_ts=400, nf2ff_freqs=freqs)
    sd.build(oracle_lib).run(400)
    sr = patch_sim(36, 34, 30, nr_ts=400, nf2ff_freqs=freqs, nf2ff_mode="record")
    er = sr.build(oracle_lib)
    er.run(400)
    assert sr.nf2ff_mode == "record" and sr.rec_bytes > 0
    for a, b in zip(sd.nf2ff_boxes(), sr.nf2ff_boxes()):
        assert np.array_equal(a, b)                 # same float64 fma chain, sample by sample
    # any other frequency afterwards == a fresh running-DFT run that had named it beforehand
    other = np.array([2.1e9])
    s2 = patch_sim(36, 34, 30, nr_ts=400, nf2ff_freqs=other)
    s2.build(oracle_lib).run(400)
    assert s2.dft_every == sr.dft_every
    for a, b in zip(s2.nf2ff_boxes(), sr.nf2ff_boxes(freqs=other)):
        assert np.array_equal(a, b)
    # auto mode falls back to the running DFT when the record would not fit the budget
    sa = pkg("simulation").Simulation(sd.grid, sd.vox, f0=sd.f0, fc=sd.fc, boundary="CPML", cpml_cells=8, nr_ts=400,
                                      nf2ff_freqs=freqs, nf2ff_mode="auto", rec_budget_bytes=1000)
    assert sa.nf2ff_mode == "dft"

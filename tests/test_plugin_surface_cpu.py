"""The drop-in boundary, pinned against the reference: every prepare_hip_* must emit exactly the call
sequence (boxes, materials, mesh hint lines, ports, NrTS, EndCriteria, f0/fc, boundary) and the
theta/phi/nf_center that the corresponding reference prepare_* emitted when it was imported against a
recording fake (tests/golden/make_fixtures.py -> scene_calls.json), and the result conversion must
reproduce the reference's dBi arithmetic on synthetic NF2FF data (result_conversion.json)."""
import json
import os

import numpy as np
import pytest

from conftest import ROOT, pkg

GOLD = os.path.join(ROOT, "tests", "golden")


def _load(name):
    with open(os.path.join(GOLD, name)) as fh:
        return json.load(fh)


def _same(a, b, path=""):
    if isinstance(a, dict) and isinstance(b, dict):
        assert sorted(a) == sorted(b), f"{path}: keys {sorted(a)} != {sorted(b)}"
        for k in a:
            _same(a[k], b[k], f"{path}.{k}")
    elif isinstance(a, (list, tuple)) and isinstance(b, (list, tuple)):
        assert len(a) == len(b), f"{path}: len {len(a)} != {len(b)}"
        for i, (x, y) in enumerate(zip(a, b)):
            _same(x, y, f"{path}[{i}]")
    elif isinstance(a, (int, float)) and isinstance(b, (int, float)) and not isinstance(a, bool):
        assert abs(a - b) <= 1e-9 * max(1.0, abs(a), abs(b)), f"{path}: {a} != {b}"
    else:
        assert a == b, f"{path}: {a!r} != {b!r}"


def _params(f=2.45e9, **kw):
    P = pkg("params").PatchAntennaParams
    return P.from_user_units(frequency_ghz=f / 1e9, er=4.3, h_mm=1.6, loss_tangent=0.02, **kw)


def _cases():
    s = pkg("solver_fdtd_hip")
    FD, PI = s.FeedDirection, s.PatchInstance
    pitch = 0.0612
    arr = [PI(name=f"P{n}", params=_params(), center_x_m=(ix - 0.5) * pitch, center_y_m=(iy - 0.5) * pitch, center_z_m=0.0,
              feed_direction=FD.NEG_X) for n, (ix, iy) in enumerate([(0, 0), (1, 0), (0, 1), (1, 1)])]
    rot = [PI(name="R1", params=_params(), center_x_m=0.0, center_y_m=0.0, center_z_m=0.01, feed_direction=FD.NEG_Y, rot_z_deg=90.0),
           PI(name="R2", params=_params(), center_x_m=0.08, center_y_m=0.0, center_z_m=0.0, feed_direction=FD.POS_X, rot_x_deg=90.0)]
    return {
        "fixed_2g45": lambda: s.prepare_hip_patch_fixed(_params()),
        "fixed_explicit_LW": lambda: s.prepare_hip_patch_fixed(_params(L_mm=28.0, W_mm=36.0)),
        "microstrip_negx": lambda: s.prepare_hip_microstrip_patch(_params(), feed_direction=FD.NEG_X, boundary="MUR", theta_step_deg=2.0),
        "microstrip_posy": lambda: s.prepare_hip_microstrip_patch(_params(), feed_direction=FD.POS_Y, boundary="MUR", theta_step_deg=2.0),
        "microstrip3d_5g8_pml_q3": lambda: s.prepare_hip_microstrip_patch_3d(_params(5.8e9), feed_direction=FD.NEG_X, boundary="PML_8",
                                                                           theta_step_deg=2.0, phi_step_deg=5.0, mesh_quality=3),
        "microstrip3d_2g45_mur_q5_posx": lambda: s.prepare_hip_microstrip_patch_3d(_params(), feed_direction=FD.POS_X, boundary="MUR",
                                                                                 theta_step_deg=5.0, phi_step_deg=10.0, mesh_quality=5),
        "multi_2x2": lambda: s.prepare_hip_microstrip_multi_3d(arr, boundary="PML_8", theta_step_deg=2.0, phi_step_deg=5.0, mesh_quality=3),
        "multi_rotated": lambda: s.prepare_hip_microstrip_multi_3d(rot, boundary="MUR", theta_step_deg=4.0, phi_step_deg=10.0, mesh_quality=6,
                                                                   nf_center_mode="centroid", end_criteria_db=-40.0),
        "multi_manual_box": lambda: s.prepare_hip_microstrip_multi_3d(arr[:1], boundary="MUR", simbox_mode="manual",
                                                                      manual_size_mm=(260.0, 240.0, 200.0), mesh_quality=2),
        "legacy_2g45": lambda: s.prepare_hip_patch(_params()),
    }


@pytest.mark.parametrize("name", sorted(_load("scene_calls.json")))
def test_prepare_emits_reference_call_sequence(name):
    gold = _load("scene_calls.json")[name]
    prep = _cases()[name]()
    assert prep.ok, prep.message
    _same(prep.FDTD.calls, gold["calls"], name)
    _same(np.asarray(prep.theta).tolist(), gold["theta"], name + ".theta")
    _same(np.asarray(prep.phi).tolist(), gold["phi"], name + ".phi")
    _same(np.asarray(prep.nf_center).tolist(), gold["nf_center"], name + ".nf_center")
    assert prep.sim_path and os.path.isabs(prep.sim_path)


def test_design_helpers_match_reference_values():
    pd = pkg("patch_design")
    for row in _load("design_values.json"):
        L, W, ee = pd.design_patch_for_frequency(row["f"], row["eps_r"], row["h"])
        for got, want in ((L, row["L"]), (W, row["W"]), (ee, row["eps_eff"]), (pd.delta_L(ee, row["h"], W), row["dL"]),
                          (pd.effective_eps(row["eps_r"], row["h"], W), row["eps_eff_direct"]),
                          (pd.calculate_microstrip_width(row["f"], row["eps_r"], row["h"]), row["w50"]),
                          (pd.calculate_microstrip_width(row["f"], row["eps_r"], row["h"], 30.0), row["w30"]),
                          (pd.calculate_microstrip_width(row["f"], row["eps_r"], row["h"], 75.0), row["w75"])):
            assert abs(got - want) <= 1e-12 * abs(want)


def test_input_model_matches_reference():
    gold = _load("input_model.json")
    P = pkg("params").PatchAntennaParams
    p = _params(metal="gold", metal_thickness_um=3.0)
    _same(json.loads(p.model_dump_json()), gold["from_user_units"])
    _same(json.loads(P(frequency_hz=1e9, eps_r=2.2, h_m=1e-3).model_dump_json()), gold["defaults"])
    _same({"frequency_ghz": p.frequency_ghz, "h_mm": p.h_mm, "L_mm": p.L_mm, "W_mm": p.W_mm}, gold["props"])
    FD = pkg("solver_fdtd_hip").FeedDirection
    assert {m.name: m.value for m in FD} == gold["feed_directions"]
    bad = []
    for kw in ({"frequency_hz": -1, "eps_r": 2, "h_m": 1e-3}, {"frequency_hz": 1e9, "eps_r": 1.0, "h_m": 1e-3},
               {"frequency_hz": 1e9, "eps_r": 2, "h_m": 0}, {"frequency_hz": 1e9, "eps_r": 2, "h_m": 1e-3, "loss_tangent": -0.1}):
        try:
            P(**kw)
            bad.append(False)
        except Exception:
            bad.append(True)
    assert bad == gold["rejects"]


@pytest.mark.parametrize("variant,key", [("fixed", "fixed"), ("microstrip", "microstrip"), ("microstrip_3d", "microstrip3d"),
                                         ("multi_3d", "multi")])
def test_result_conversion_matches_reference(variant, key):
    """Same synthetic NF2FF data the fixture generator fed the reference -> same dBi grid."""
    gold = _load("result_conversion.json")[key]
    s = pkg("solver_fdtd_hip")
    th = np.asarray(gold["theta"])[:, None]
    ph = np.asarray(gold["phi"])[None, :]
    E = np.cos(th / 2.0) ** 2 * (1.0 + 0.25 * np.cos(ph)) + 1e-3
    got = s.pattern_to_dBi(E, 4.0, variant)
    assert got.shape == np.asarray(gold["intensity"]).shape
    assert np.allclose(got, np.asarray(gold["intensity"]), rtol=0, atol=1e-9)
    assert gold["is_dBi"] is True


@pytest.mark.parametrize("key,mode", [("legacy", "full"), ("legacy_enorm_fallback", "weak"), ("legacy_fields_only", "fields")])
def test_legacy_result_conversion_matches_reference(key, mode):
    """The legacy variant (solver_fdtd_openems.py:271-411): radian theta / phi in Prepared, DEGREES handed to CalcNF2FF,
    radians out; its three conversion branches — directivity from P_rad / Prad, E_norm + Dmax when that looks wrong,
    normalised |E|^2 (linear, is_dBi False) when only the field components exist — against what the reference made of
    the same synthetic far field (tests/golden/make_fixtures.py: legacy_synthetic)."""
    import types
    gold = _load("result_conversion.json")[key]
    s = pkg("solver_fdtd_hip")
    prep = s.prepare_hip_patch(_params())
    assert prep.ok and prep.variant == "legacy"
    _same(np.asarray(prep.theta).tolist(), gold["theta"], "theta")        # radians already in Prepared ...
    _same(np.asarray(prep.phi).tolist(), gold["phi"], "phi")              # ... and unchanged in the result
    th, ph = np.asarray(prep.theta)[:, None], np.asarray(prep.phi)[None, :]
    e_th = np.cos(th / 2.0) ** 2 * (1.0 + 0.25 * np.cos(ph)) * np.exp(0.3j) + 0j
    e_ph = 0.3 * np.sin(th) * np.sin(ph) * np.exp(-0.7j) + 0j
    p_rad = (np.abs(e_th) ** 2 + np.abs(e_ph) ** 2) / (2.0 * 376.730313668)
    prad = float(np.sum(p_rad * np.sin(th)) * (th[1, 0] - th[0, 0]) * (ph[0, 1] - ph[0, 0]))
    res = types.SimpleNamespace()
    if mode in ("full", "weak"):
        res.E_theta, res.E_phi = [e_th], [e_ph]
        res.E_norm = [np.sqrt(np.abs(e_th) ** 2 + np.abs(e_ph) ** 2)]
        res.P_rad = [p_rad * (1e-4 if mode == "weak" else 1.0)]
        res.Prad, res.Dmax = [prad], [float(4.0 * np.pi * p_rad.max() / prad)]
    else:
        res.E_theta, res.E_phi = e_th, e_ph
    got, dbi = s.legacy_pattern(res, th.size, ph.size)
    assert dbi is gold["is_dBi"] and list(got.shape) == gold["shape"]
    sub = gold["sub"]
    assert np.allclose(got[::sub[0], ::sub[1]], np.asarray(gold["intensity_sub"]), rtol=0, atol=1e-9)
    assert abs(got.sum() - gold["sum"]) <= 1e-9 * max(1.0, abs(gold["sum"])) and abs((got * got).sum() - gold["sum_sq"]) <= 1e-9 * gold["sum_sq"]
    # the reference hands DEGREES to CalcNF2FF (openems.py:299-300)
    c = gold["calc_calls"][0]
    assert c["ntheta"] == 91 and c["nphi"] == 181 and c["theta_first_last"] == [0.0, 180.0] and c["phi_first_last"] == [0.0, 360.0]
    assert c["center"] == [0.0, 0.0, 0.001]


def test_legacy_run_hands_degrees_to_calcnf2ff_and_returns_radians():
    """run_prepared_hip on a legacy Prepared whose engine objects are recording fakes: same CalcNF2FF arguments as the
    reference's (fixture above), result theta / phi in radians, P_rad branch -> dBi."""
    import types
    s = pkg("solver_fdtd_hip")
    prep = s.prepare_hip_patch(_params())
    seen = {}

    class FakeNF:
        def can_evaluate(self, f):
            return True

        def CalcNF2FF(self, sim_path, freq, theta, phi, center=None, **kw):
            seen.update(freq=freq, theta=np.asarray(theta), phi=np.asarray(phi), center=np.asarray(center))
            th, ph = np.deg2rad(theta)[:, None], np.deg2rad(phi)[None, :]
            e = np.cos(th / 2.0) ** 2 * (1.0 + 0.25 * np.cos(ph)) + 1e-3
            p = e ** 2
            prad = float(np.sum(p * np.sin(th)) * (th[1, 0] - th[0, 0]) * (ph[0, 1] - ph[0, 0]))
            return types.SimpleNamespace(E_norm=[e], Dmax=[4 * np.pi * p.max() / prad], P_rad=[p], Prad=[prad], freq=[freq])

    class FakeFDTD:
        stats = types.SimpleNamespace(steps=1, seconds=1.0, mcells_per_s=1.0, energy_db=-60.0)
        sim = types.SimpleNamespace(dt=1e-12, grid=types.SimpleNamespace(ncells=8, shape=(2, 2, 2)))

        def Run(self, sim_path, verbose=0, cleanup=False):
            seen["run"] = (verbose, cleanup)

    prep.FDTD, prep.nf, prep.port = FakeFDTD(), FakeNF(), None
    r = s.run_prepared_hip(prep, frequency_hz=2.45e9, verbose=0)
    assert r.ok, r.message
    assert seen["theta"][0] == 0.0 and seen["theta"][-1] == 180.0 and seen["phi"][-1] == 360.0 and seen["theta"].size == 91
    assert np.allclose(seen["center"], [0, 0, 1e-3]) and seen["freq"] == 2.45e9 and seen["run"] == (0, False)
    assert r.theta[-1] == np.pi and r.phi[-1] == 2 * np.pi and r.intensity.shape == (91, 181) and r.is_dBi
    assert abs(r.intensity.max() - 10 * np.log10(r.Dmax)) < 1e-9


def test_never_raises_returns_ok_false():
    s = pkg("solver_fdtd_hip")
    r = s.prepare_hip_microstrip_multi_3d([])
    assert not r.ok and "No patch" in r.message
    bad = s.prepare_hip_patch_fixed(_params(), dll_dir="/nonexistent/dir")
    assert not bad.ok and "prepare failed" in bad.message
    res = s.run_prepared_hip(bad, frequency_hz=2.45e9, verbose=0)
    assert not res.ok and res.message == bad.message
    pr = s.probe_hip("/nonexistent/dir")
    assert not pr.ok


@pytest.mark.parametrize("f_ghz,er,h_mm", [(0.9, 4.3, 0.254), (5.8, 10.2, 3.2), (2.45, 2.2, 0.8)])
def test_every_variant_sets_up_over_a_range_of_designs(oracle_lib, tmp_path, f_ghz, er, h_mm):
    """Substrates from a hundredth to a fifteenth of the mesh resolution, permittivities 2.2 ... 10.2: every variant prepares and sets its
    engine up — mesh, voxels, operator, a lumped port with a non-zero length (five lines 63 um apart across a 0.254 mm substrate under an
    11 mm mesh must survive the merging of accidentally close hint lines), NF2FF box — and says so when the pulse outlasts the reference's
    NrTS (0.9 GHz on 0.254 mm: 64 um cells, the pulse needs 31 000 of the 30 000 timesteps; openEMS prints its own warning there)."""
    import warnings
    s = pkg("solver_fdtd_hip")
    p = pkg("params").PatchAntennaParams.from_user_units(frequency_ghz=f_ghz, er=er, h_mm=h_mm, loss_tangent=0.002)
    for n, prep_fn in enumerate((s.prepare_hip_patch_fixed, s.prepare_hip_microstrip_patch, s.prepare_hip_microstrip_patch_3d, s.prepare_hip_patch)):
        prep = prep_fn(p, work_dir=str(tmp_path / f"v{n}"), lib=oracle_lib)
        assert prep.ok, prep.message
        with warnings.catch_warnings(record=True):
            warnings.simplefilter("always")
            prep.FDTD.Run(prep.sim_path, verbose=0, cleanup=False, setup_only=True)
        sim = prep.FDTD.sim
        assert min(float(np.min(np.diff(l))) for l in sim.grid.lines) > 5e-6 and sim.grid.ncells < 5e6
        assert (sim.excitation_warning is not None) == (len(sim.signal) > sim.nr_ts)
        if (f_ghz, h_mm) == (0.9, 0.254) and prep_fn is not s.prepare_hip_patch:
            assert sim.excitation_warning and "ends before the pulse" in sim.excitation_warning
        sim.engine.close()

"""The double-precision build of the oracle (oracle/Makefile: libfdtd_oracle_f64.so, -DFDTD_REAL=double) against the float32 oracle — the
checker the HIP library equals bit for bit — on small scenes: what float32 costs over a run stays an order inside north_star's 1e-3.
The full-length record (the reference's default scene to -40 dB, the north-star grid for 12 000 timesteps): tests/fp32_error_budget.py ->
profiles/r04/fp32_error_budget.json."""
import numpy as np
import pytest

from conftest import pkg
from helpers import patch_sim, rel_l2, load_oracle_f64, build_f64


@pytest.fixture(scope="module")
def oracle_f64(oracle_lib):      # (oracle_lib builds oracle/ on demand: both libraries)
    return load_oracle_f64()


def test_double_build_exports_the_abi_and_says_what_it_is(oracle_lib, oracle_f64):
    capi = pkg("_capi")
    for name in capi.ABI_SYMBOLS:
        assert hasattr(oracle_f64, name), name
    assert oracle_f64.fdtd_oracle_real_bytes() == 8
    oracle_lib.fdtd_oracle_real_bytes.restype = int
    assert oracle_lib.fdtd_oracle_real_bytes() == 4
    # staging double tables is refused by the float build
    import ctypes
    a = np.ones(4)
    oracle_lib.fdtd_oracle_stage_f64.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t]
    assert oracle_lib.fdtd_oracle_stage_f64(0, a.ctypes.data, a.size) == -5


@pytest.mark.parametrize("boundary", ["CPML", "MUR"])
def test_float32_run_within_1e4_of_the_double_run(oracle_lib, oracle_f64, boundary):
    steps = 2500
    runs = {}
    for tag, lib, dbl in (("f32", oracle_lib, None), ("f64_f32tables", oracle_f64, False), ("f64", oracle_f64, True)):
        s = patch_sim(44, 40, 30, boundary=boundary, cpml_cells=8, nr_ts=steps)
        e = s.build(lib) if dbl is None else build_f64(s, lib, double_tables=dbl)
        e.run(steps)
        u, i = s.port_series()[0]
        runs[tag] = (u, i, e.fields(), s.nf2ff_boxes())
    for ref in ("f64_f32tables", "f64"):
        u32, i32, f32, b32 = runs["f32"]
        u, i, f, b = runs[ref]
        assert np.abs(u).max() > 0
        assert rel_l2(u32, u) < 1e-4 and rel_l2(i32, i) < 1e-4, (ref, rel_l2(u32, u), rel_l2(i32, i))
        assert rel_l2(f32, f) < 1e-4, (ref, rel_l2(f32, f))
        assert max(rel_l2(x, y) for x, y in zip(b32, b)) < 1e-4
    # the two double runs differ by the rounding of the coefficients only — and differ they must (the staged tables were taken)
    d = rel_l2(runs["f64_f32tables"][0], runs["f64"][0])
    assert 0 < d < 1e-5, d

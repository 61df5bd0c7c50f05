"""Round 4, GPU: launches that cannot abort the process (occupancy caps sized from the kernels' own LDS), and other robustness checks."""
import numpy as np
import pytest

from conftest import pkg
from helpers import patch_sim, seeded_fields

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("boundary", ["PEC", "CPML"])
@pytest.mark.parametrize("knob", ["FDTD_OCC_WF", "FDTD_OCC_E", "FDTD_OCC_H"])
@pytest.mark.parametrize("cap", ["1", "3"])
def test_occupancy_caps_run_or_return_an_error_code(hip_lib, monkeypatch, boundary, knob, cap):
    """An occupancy cap pads the dynamic LDS of a launch; round 3 sized the padding from hand-kept estimates of the kernels' static LDS and a cap
    of 1 on a kernel without CPML staging asked for more than a workgroup may have — the runtime ABORTED the process.  The padding now comes
    from hipFuncGetAttributes and the device's limits: every cap either runs (same fields as without it) or comes back as an error code."""
    capi = pkg("_capi")
    flags = capi.FLAG_KERNEL_DIRECT if knob != "FDTD_OCC_WF" else capi.FLAG_KERNEL_WAVEFRONT

    def run(env):
        if env:
            monkeypatch.setenv(knob, cap)
        s = patch_sim(60, 52, 34, boundary=boundary, cpml_cells=6, nr_ts=60, nf2ff=False)
        e = s.build(hip_lib, flags=flags)
        if env:
            monkeypatch.delenv(knob)
        seeded_fields(e, 3)
        e.run(60)
        return e.fields()
    ref = run(False)
    try:
        got = run(True)
    except capi.FdtdError as exc:          # an error code with a message is a legitimate outcome; an abort is not
        assert "LDS" in str(exc) or "launch" in str(exc)
        return
    assert np.array_equal(ref, got)

"""CPU checks of the drop-in boundary: both libraries export every symbol include/fdtd_hip.h
declares; the HIP one loads without a GPU (no compute calls here)."""
import ctypes
import os
import re

from conftest import ROOT, pkg


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "fdtd_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fdtd_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree():
    capi = pkg("_capi")
    assert _declared_symbols() == sorted(capi.ABI_SYMBOLS)


def test_hip_library_exports_abi():
    capi = pkg("_capi")
    path = capi.hip_library_path()
    assert os.path.isfile(path), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(path)
    for s in _declared_symbols():
        assert hasattr(lib, s), s
    capi.bind(lib)
    assert lib.fdtd_version() == 4
    assert lib.fdtd_backend() == b"hip:gfx950"


def test_oracle_exports_abi(oracle_lib):
    for s in _declared_symbols():
        assert hasattr(oracle_lib, s), s
    assert oracle_lib.fdtd_backend() == b"oracle:cpu"


def test_missing_library_fails_loudly(tmp_path):
    capi = pkg("_capi")
    import pytest
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        capi.load_hip_library(str(tmp_path))


def test_create_without_gpu_reports_error():
    """On a GPU-less host fdtd_create must fail with a message, not crash or fall back."""
    capi = pkg("_capi")
    lib = capi.bind(ctypes.CDLL(capi.hip_library_path()))
    if lib.fdtd_device_count() > 0:
        return
    import pytest
    with pytest.raises(capi.FdtdError, match="no HIP device"):
        capi.Engine(lib, 8, 8, 8, 1e-12)


def test_flag_constants_of_the_binding_match_the_header():
    """The ctypes stub spells the kernel-schedule / transport flags as literals: they must be the header's values."""
    import re
    capi = pkg("_capi")
    text = open(os.path.join(ROOT, "include", "fdtd_hip.h")).read()
    enum = dict((m.group(1), int(m.group(2), 0)) for m in re.finditer(r"\b(FDTD_FLAG_\w+)\s*=\s*(0x[0-9A-Fa-f]+|\d+)", text))
    for name in ("KERNEL_AUTO", "KERNEL_DIRECT", "KERNEL_WAVEFRONT", "KERNEL_MASK", "OVERLAP_ON", "OVERLAP_OFF", "LOOPBACK"):
        assert getattr(capi, "FLAG_" + name) == enum["FDTD_FLAG_" + name], name

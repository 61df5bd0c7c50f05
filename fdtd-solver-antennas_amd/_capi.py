"""ctypes binding of the C ABI in include/fdtd_hip.h.

``load_hip_library()`` is the ONLY loader the product uses; it raises if libfdtd_hip.so is
missing (there is no CPU fallback).  ``bind()`` can attach the same prototypes to any library
exporting the ABI — tests use that to drive oracle/libfdtd_oracle.so as the checker.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_LIB_NAME = "libfdtd_hip.so"

ABI_SYMBOLS = [
    "fdtd_version", "fdtd_device_count", "fdtd_backend", "fdtd_create", "fdtd_destroy",
    "fdtd_last_error", "fdtd_set_operator_raw", "fdtd_set_operator_classes", "fdtd_build_operator",
    "fdtd_operator_form", "fdtd_get_operator", "fdtd_set_cpml",
    "fdtd_set_mur", "fdtd_set_signal", "fdtd_add_source", "fdtd_add_probe", "fdtd_get_probe",
    "fdtd_set_dft", "fdtd_add_dft_box", "fdtd_get_dft_box", "fdtd_set_recorder", "fdtd_rec_transform", "fdtd_run", "fdtd_run_profiled",
    "fdtd_get_step", "fdtd_energy", "fdtd_p2p_export", "fdtd_p2p_attach", "fdtd_p2p_selftest", "fdtd_p2p_detach", "fdtd_p2p_link_info", "fdtd_schedule_info", "fdtd_comm_unique_id", "fdtd_comm_init", "fdtd_comm_nranks", "fdtd_link", "fdtd_run_linked", "fdtd_half_step",
    "fdtd_halo_get", "fdtd_halo_put", "fdtd_get_field", "fdtd_set_field", "fdtd_farfield",
]


class FdtdDesc(C.Structure):
    _fields_ = [("nx", C.c_int32), ("ny", C.c_int32), ("nz", C.c_int32),
                ("k0", C.c_int32), ("nk", C.c_int32),
                ("rank", C.c_int32), ("world", C.c_int32),
                ("device", C.c_int32), ("max_steps", C.c_int32),
                ("flags", C.c_uint32), ("dt", C.c_double)]


class FdtdProfile(C.Structure):
    _fields_ = [("ms_total", C.c_double), ("ms_update_e", C.c_double), ("ms_update_h", C.c_double),
                ("launches_e", C.c_int32), ("launches_h", C.c_int32),
                ("steps", C.c_int32), ("fused", C.c_int32), ("ms_event_overhead", C.c_double)]


FLAG_KERNEL_AUTO, FLAG_KERNEL_DIRECT, FLAG_KERNEL_WAVEFRONT, FLAG_KERNEL_RESIDENT, FLAG_KERNEL_MASK = 0, 1, 5, 6, 0xF
FLAG_OVERLAP_ON, FLAG_OVERLAP_OFF, FLAG_LOOPBACK = 0x20, 0x40, 0x80
KIND_V, KIND_I = 0, 1
PHASE_E, PHASE_H = 0, 1
HALO_H_UP, HALO_E_DOWN = 0, 1

_vp, _i, _i64p = C.c_void_p, C.c_int, C.POINTER(C.c_int64)


def bind(lib: C.CDLL) -> C.CDLL:
    """Attach argtypes/restype for every ABI symbol (raises AttributeError if one is missing)."""
    p = C.c_void_p
    sig = {
        "fdtd_version": (C.c_int, []),
        "fdtd_device_count": (C.c_int, []),
        "fdtd_backend": (C.c_char_p, []),
        "fdtd_create": (C.c_int, [C.POINTER(FdtdDesc), C.POINTER(p)]),
        "fdtd_destroy": (None, [p]),
        "fdtd_last_error": (C.c_char_p, [p]),
        "fdtd_set_operator_raw": (C.c_int, [p, p, p, p, p]),
        "fdtd_set_operator_classes": (C.c_int, [p, p, C.c_int, p, p, p, p]),
        "fdtd_build_operator": (C.c_int, [p, p, p, p, p, p, p, C.c_double, C.c_int, p, p, p, p, p, p, C.c_int]),
        "fdtd_operator_form": (C.c_int, [p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "fdtd_get_operator": (C.c_int, [p, p, p, p, p]),
        "fdtd_set_cpml": (C.c_int, [p, p, p, p, C.c_int, C.c_int, C.c_int, p]),
        "fdtd_set_mur": (C.c_int, [p, p, p]),
        "fdtd_set_signal": (C.c_int, [p, p, C.c_int]),
        "fdtd_add_source": (C.c_int, [p, C.c_int, p, p, p, p]),
        "fdtd_add_probe": (C.c_int, [p, C.c_int, C.c_int, p, p, p, C.POINTER(C.c_int)]),
        "fdtd_get_probe": (C.c_int, [p, C.c_int, p, C.c_int, C.POINTER(C.c_int)]),
        "fdtd_set_dft": (C.c_int, [p, C.c_int, C.c_int, C.c_int, p, p]),
        "fdtd_add_dft_box": (C.c_int, [p, C.c_int, C.c_int, p, p, C.POINTER(C.c_int)]),
        "fdtd_get_dft_box": (C.c_int, [p, C.c_int, p, p, p]),
        "fdtd_set_recorder": (C.c_int, [p, C.c_int, C.c_int]),
        "fdtd_rec_transform": (C.c_int, [p, C.c_int, C.c_int, p, p, p, p]),
        "fdtd_run": (C.c_int, [p, C.c_int]),
        "fdtd_run_profiled": (C.c_int, [p, C.c_int, C.POINTER(FdtdProfile)]),
        "fdtd_get_step": (C.c_int, [p, C.POINTER(C.c_int64)]),
        "fdtd_energy": (C.c_int, [p, p]),
        "fdtd_p2p_export": (C.c_int, [p, p]),
        "fdtd_p2p_attach": (C.c_int, [p, p, p]),
        "fdtd_p2p_selftest": (C.c_int, [p, C.c_uint]),
        "fdtd_p2p_detach": (C.c_int, [p]),
        "fdtd_p2p_link_info": (C.c_int, [p, C.c_int, p]),
        "fdtd_schedule_info": (C.c_int, [p, p]),
        "fdtd_comm_unique_id": (C.c_int, [p]),
        "fdtd_comm_init": (C.c_int, [p, p]),
        "fdtd_comm_nranks": (C.c_int, [p, C.POINTER(C.c_int)]),
        "fdtd_link": (C.c_int, [p, p]),
        "fdtd_run_linked": (C.c_int, [C.POINTER(p), C.c_int, C.c_int]),
        "fdtd_half_step": (C.c_int, [p, C.c_int]),
        "fdtd_halo_get": (C.c_int, [p, C.c_int, p]),
        "fdtd_halo_put": (C.c_int, [p, C.c_int, p]),
        "fdtd_get_field": (C.c_int, [p, C.c_int, C.c_int, p]),
        "fdtd_set_field": (C.c_int, [p, C.c_int, C.c_int, p]),
        "fdtd_farfield": (C.c_int, [C.c_int, C.c_int, p, p, p, C.c_double, C.c_int, p, p, p, p]),
    }
    assert sorted(sig) == sorted(ABI_SYMBOLS)
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


def hip_library_path(lib_dir: Optional[str] = None) -> str:
    """<lib_dir>/libfdtd_hip.so; default: $FDTD_HIP_LIB_DIR, else the in-tree build under csrc/."""
    return os.path.join(lib_dir or os.environ.get("FDTD_HIP_LIB_DIR") or os.path.join(_HERE, "csrc"), HIP_LIB_NAME)


_hip_lib = None


def load_hip_library(lib_dir: Optional[str] = None) -> C.CDLL:
    """Load libfdtd_hip.so (built by __graft_entry__.build() / csrc/Makefile).  No fallback."""
    global _hip_lib
    if _hip_lib is not None and lib_dir is None:
        return _hip_lib
    path = hip_library_path(lib_dir)
    if not os.path.isfile(path):
        raise RuntimeError(
            f"{HIP_LIB_NAME} not found at {path}: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the FDTD hot path.")
    lib = bind(C.CDLL(path))
    if lib_dir is None:
        _hip_lib = lib
    return lib


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _arr(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


class FdtdError(RuntimeError):
    pass


class Engine:
    """Thin object wrapper over one fdtd_ctx (one z-slab on one device)."""

    def __init__(self, lib: C.CDLL, nx: int, ny: int, nz: int, dt: float, *, k0: int = 0,
                 nk: Optional[int] = None, rank: int = 0, world: int = 1, device: int = 0,
                 max_steps: int = 0, flags: int = 0):
        self.lib = lib
        nk = nz - k0 if nk is None else nk
        self.desc = FdtdDesc(nx, ny, nz, k0, nk, rank, world, device, max_steps, flags, dt)
        self._ctx = C.c_void_p()
        rc = lib.fdtd_create(C.byref(self.desc), C.byref(self._ctx))
        if rc != 0:
            msg = lib.fdtd_last_error(None)
            raise FdtdError(f"fdtd_create failed ({rc}): {msg.decode() if msg else ''}")
        self.nx, self.ny, self.nz, self.k0, self.nk = nx, ny, nz, k0, nk
        self._keep = []

    # -- plumbing ---------------------------------------------------------------------------
    def _ck(self, rc: int, what: str):
        if rc != 0:
            msg = self.lib.fdtd_last_error(self._ctx)
            raise FdtdError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    def close(self):
        if self._ctx:
            self.lib.fdtd_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def backend(self) -> str:
        return self.lib.fdtd_backend().decode()

    @property
    def local_shape(self):
        return (self.nk, self.ny, self.nx)

    # -- operator -----------------------------------------------------------------------------
    def set_operator_raw(self, vv, vi, ii, iv):
        shp = (3,) + self.local_shape
        arrs = [_arr(a, np.float32) for a in (vv, vi, ii, iv)]
        for a in arrs:
            if a.shape != shp:
                raise ValueError(f"operator array shape {a.shape} != {shp}")
        self._ck(self.lib.fdtd_set_operator_raw(self._ctx, *[_ptr(a) for a in arrs]), "set_operator_raw")

    def set_operator_classes(self, ecls, cls_vv, cls_m, emet, hmet):
        shp = (3,) + self.local_shape
        ecls = _arr(ecls, np.uint8)
        if ecls.shape != shp:
            raise ValueError(f"class array shape {ecls.shape} != {shp}")
        cls_vv, cls_m = _arr(cls_vv, np.float32), _arr(cls_m, np.float32)
        emet, hmet = _arr(emet, np.float32), _arr(hmet, np.float32)
        tl = self.nx + self.ny + self.nk
        if emet.shape != (3, tl) or hmet.shape != (3, tl) or cls_vv.shape != cls_m.shape:
            raise ValueError("metric/class table shape mismatch")
        self._ck(self.lib.fdtd_set_operator_classes(self._ctx, _ptr(ecls), int(cls_vv.size), _ptr(cls_vv),
                                                    _ptr(cls_m), _ptr(emet), _ptr(hmet)), "set_operator_classes")

    def build_operator(self, d, eps_r, kappa, pec, eps0, overrides, emet, hmet, prefer_classes=True):
        """Operator set-up inside the library (on the GPU for libfdtd_hip.so).  d = (dx, dy, dz) primal edge
        lengths of the GLOBAL grid; eps_r/kappa per cell [nz-1][ny-1][nx-1]; pec [3][nz][ny][nx];
        overrides = (edge_index int64, comp int8, vv float32, m float32) for host-fixed (lumped) edges."""
        nx, ny, nz = self.nx, self.ny, self.nz
        dx, dy, dz = (_arr(a, np.float64) for a in d)
        if (dx.size, dy.size, dz.size) != (nx, ny, nz):
            raise ValueError("cell-size tables must have nx, ny, nz entries")
        eps_r, kappa = _arr(eps_r, np.float64), _arr(kappa, np.float64)
        if eps_r.shape != (nz - 1, ny - 1, nx - 1) or kappa.shape != eps_r.shape:
            raise ValueError("cell arrays must be [nz-1][ny-1][nx-1]")
        pec = np.ascontiguousarray(pec)
        if pec.dtype == np.bool_:
            pec = pec.view(np.uint8)
        pec = _arr(pec, np.uint8)
        if pec.shape != (3, nz, ny, nx):
            raise ValueError("pec must be [3][nz][ny][nx]")
        oe, oc, ov, om = overrides
        oe, oc, ov, om = _arr(oe, np.int64), _arr(oc, np.int8), _arr(ov, np.float32), _arr(om, np.float32)
        if not (oe.size == oc.size == ov.size == om.size):
            raise ValueError("override arrays differ in length")
        emet, hmet = _arr(emet, np.float32), _arr(hmet, np.float32)
        tl = nx + ny + self.nk
        if emet.shape != (3, tl) or hmet.shape != (3, tl):
            raise ValueError("metric table shape mismatch")
        self._ck(self.lib.fdtd_build_operator(self._ctx, _ptr(dx), _ptr(dy), _ptr(dz), _ptr(eps_r), _ptr(kappa), _ptr(pec),
                                              float(eps0), int(oe.size), _ptr(oe), _ptr(oc), _ptr(ov), _ptr(om),
                                              _ptr(emet), _ptr(hmet), 1 if prefer_classes else 0), "build_operator")

    def operator_form(self):
        """('none' | 'classes' | 'classes-packed' | 'raw', number of distinct (vv, m) pairs)."""
        form, ncls = C.c_int(0), C.c_int(0)
        self._ck(self.lib.fdtd_operator_form(self._ctx, C.byref(form), C.byref(ncls)), "operator_form")
        return ("none", "classes", "classes-packed", "raw")[form.value], int(ncls.value)

    def get_operator(self):
        """(vv, vi, ii, iv) float32 [3][nk][ny][nx]: the operator that is set, expanded."""
        out = [np.empty((3,) + self.local_shape, np.float32) for _ in range(4)]
        self._ck(self.lib.fdtd_get_operator(self._ctx, *[_ptr(a) for a in out]), "get_operator")
        return tuple(out)

    # -- boundaries ---------------------------------------------------------------------------
    def set_cpml(self, slot_x, slot_y, slot_z, nsx, nsy, nsz, coef):
        sx, sy, sz = _arr(slot_x, np.int32), _arr(slot_y, np.int32), _arr(slot_z, np.int32)
        coef = _arr(coef, np.float32)
        if sx.size != self.nx or sy.size != self.ny or sz.size != self.nk:
            raise ValueError("cpml slot table length mismatch")
        if coef.size != 6 * (self.nx + self.ny + self.nk):
            raise ValueError("cpml coefficient block size mismatch")
        self._ck(self.lib.fdtd_set_cpml(self._ctx, _ptr(sx), _ptr(sy), _ptr(sz), int(nsx), int(nsy), int(nsz),
                                        _ptr(coef)), "set_cpml")

    def set_mur(self, enable, coeff):
        en, co = _arr(enable, np.int32), _arr(coeff, np.float32)
        if en.size != 6 or co.size != 6:
            raise ValueError("mur needs 6 faces")
        self._ck(self.lib.fdtd_set_mur(self._ctx, _ptr(en), _ptr(co)), "set_mur")

    # -- excitation / probes / dft -----------------------------------------------------------------
    def set_signal(self, sig):
        sig = _arr(sig, np.float32)
        self._ck(self.lib.fdtd_set_signal(self._ctx, _ptr(sig), int(sig.size)), "set_signal")

    def add_source(self, idx, comp, amp, delay=None):
        idx = _arr(idx, np.int64)
        comp = _arr(comp, np.int8)
        amp = _arr(amp, np.float32)
        delay = np.zeros(idx.size, np.int32) if delay is None else _arr(delay, np.int32)
        if not (idx.size == comp.size == amp.size == delay.size):
            raise ValueError("source arrays differ in length")
        self._ck(self.lib.fdtd_add_source(self._ctx, int(idx.size), _ptr(idx), _ptr(comp), _ptr(amp), _ptr(delay)),
                 "add_source")

    def add_probe(self, kind, idx, comp, w) -> int:
        idx, comp, w = _arr(idx, np.int64), _arr(comp, np.int8), _arr(w, np.float32)
        if not (idx.size == comp.size == w.size):
            raise ValueError("probe arrays differ in length")
        pid = C.c_int(-1)
        self._ck(self.lib.fdtd_add_probe(self._ctx, int(kind), int(idx.size), _ptr(idx), _ptr(comp), _ptr(w),
                                         C.byref(pid)), "add_probe")
        return pid.value

    def get_probe(self, pid: int) -> np.ndarray:
        n = C.c_int(0)
        self._ck(self.lib.fdtd_get_probe(self._ctx, pid, None, 0, C.byref(n)), "get_probe")
        out = np.zeros(max(n.value, 1), np.float64)
        self._ck(self.lib.fdtd_get_probe(self._ctx, pid, _ptr(out), n.value, C.byref(n)), "get_probe")
        return out[:n.value]

    def set_dft(self, every: int, tw_v: np.ndarray, tw_i: np.ndarray):
        tw_v, tw_i = _arr(tw_v, np.float64), _arr(tw_i, np.float64)
        if tw_v.shape != tw_i.shape or tw_v.ndim != 3 or tw_v.shape[2] != 2:
            raise ValueError("twiddles must be [nsamples][nfreq][2]")
        self.nfreq = tw_v.shape[1]
        self._ck(self.lib.fdtd_set_dft(self._ctx, tw_v.shape[1], int(every), tw_v.shape[0], _ptr(tw_v), _ptr(tw_i)),
                 "set_dft")

    def set_recorder(self, every: int, nsamples: int):
        """Boxes keep their time-domain samples (float32) on the device instead of running-DFT sums; `rec_transform`
        turns them into any frequency set afterwards."""
        self.nfreq = 0
        self.rec_every, self.rec_nsamples = int(every), int(nsamples)
        self._ck(self.lib.fdtd_set_recorder(self._ctx, int(every), int(nsamples)), "set_recorder")

    def rec_transform(self, bid: int, tw: np.ndarray):
        """(complex128 [nfreq][kk][jj][ii], lo_own, hi_own) of a recorded box for the twiddle table tw
        [nsamples][nfreq][2] of the box's field kind."""
        tw = _arr(tw, np.float64)
        if tw.ndim != 3 or tw.shape[2] != 2 or tw.shape[0] != self.rec_nsamples:
            raise ValueError("twiddles must be [nsamples][nfreq][2]")
        nf = tw.shape[1]
        lo, hi = np.zeros(3, np.int32), np.zeros(3, np.int32)
        self._ck(self.lib.fdtd_rec_transform(self._ctx, bid, nf, None, None, _ptr(lo), _ptr(hi)), "rec_transform")
        ext = hi - lo + 1
        if np.any(ext <= 0):
            return np.zeros((nf, 0, 0, 0), np.complex128), lo, hi
        out = np.zeros((nf, ext[2], ext[1], ext[0], 2), np.float64)
        self._ck(self.lib.fdtd_rec_transform(self._ctx, bid, nf, _ptr(tw), _ptr(out), _ptr(lo), _ptr(hi)), "rec_transform")
        return out[..., 0] + 1j * out[..., 1], lo, hi

    def add_dft_box(self, kind, comp, lo, hi) -> int:
        lo, hi = _arr(lo, np.int32), _arr(hi, np.int32)
        bid = C.c_int(-1)
        self._ck(self.lib.fdtd_add_dft_box(self._ctx, int(kind), int(comp), _ptr(lo), _ptr(hi), C.byref(bid)),
                 "add_dft_box")
        return bid.value

    def get_dft_box(self, bid: int):
        """(complex128 [nfreq][kk][jj][ii], lo_own, hi_own); the array is empty if the slab owns nothing."""
        lo, hi = np.zeros(3, np.int32), np.zeros(3, np.int32)
        self._ck(self.lib.fdtd_get_dft_box(self._ctx, bid, None, _ptr(lo), _ptr(hi)), "get_dft_box")
        ext = hi - lo + 1
        if np.any(ext <= 0):
            return np.zeros((self.nfreq, 0, 0, 0), np.complex128), lo, hi
        out = np.zeros((self.nfreq, ext[2], ext[1], ext[0], 2), np.float64)
        self._ck(self.lib.fdtd_get_dft_box(self._ctx, bid, _ptr(out), _ptr(lo), _ptr(hi)), "get_dft_box")
        return out[..., 0] + 1j * out[..., 1], lo, hi

    # -- stepping -----------------------------------------------------------------------------
    def run(self, nsteps: int):
        self._ck(self.lib.fdtd_run(self._ctx, int(nsteps)), "run")

    def run_profiled(self, nsteps: int) -> FdtdProfile:
        prof = FdtdProfile()
        self._ck(self.lib.fdtd_run_profiled(self._ctx, int(nsteps), C.byref(prof)), "run_profiled")
        return prof

    @property
    def step(self) -> int:
        s = C.c_int64(0)
        self._ck(self.lib.fdtd_get_step(self._ctx, C.byref(s)), "get_step")
        return s.value

    def energy(self):
        s = np.zeros(2, np.float64)
        self._ck(self.lib.fdtd_energy(self._ctx, _ptr(s)), "energy")
        return float(s[0]), float(s[1])

    def half_step(self, phase: int):
        self._ck(self.lib.fdtd_half_step(self._ctx, int(phase)), "half_step")

    def halo_get(self, which: int) -> np.ndarray:
        buf = np.empty((2, self.ny, self.nx), np.float32)
        self._ck(self.lib.fdtd_halo_get(self._ctx, int(which), _ptr(buf)), "halo_get")
        return buf

    def halo_put(self, which: int, buf: np.ndarray):
        buf = _arr(buf, np.float32)
        if buf.shape != (2, self.ny, self.nx):
            raise ValueError("halo buffer must be [2][ny][nx]")
        self._ck(self.lib.fdtd_halo_put(self._ctx, int(which), _ptr(buf)), "halo_put")

    def p2p_export(self) -> bytes:
        """128-byte description of this context's halo mailbox (IPC handle) for the neighbour ranks."""
        buf = C.create_string_buffer(128)
        self._ck(self.lib.fdtd_p2p_export(self._ctx, buf), "p2p_export")
        return buf.raw

    def p2p_attach(self, lower: Optional[bytes], upper: Optional[bytes]):
        """Attach the mailboxes of rank-1 / rank+1 (None where there is no neighbour): halos then travel inside the
        update kernels (csrc/kernels.hip, P2P variants)."""
        lo = C.create_string_buffer(lower, 128) if lower is not None else None
        hi = C.create_string_buffer(upper, 128) if upper is not None else None
        self._ck(self.lib.fdtd_p2p_attach(self._ctx, lo, hi), "p2p_attach")

    def p2p_selftest(self, token: int = 0x5E1F0001):
        self._ck(self.lib.fdtd_p2p_selftest(self._ctx, int(token)), "p2p_selftest")

    def p2p_detach(self):
        self._ck(self.lib.fdtd_p2p_detach(self._ctx), "p2p_detach")

    LINK_TYPES = {0: "hypertransport", 1: "qpi", 2: "pcie", 3: "infiniband", 4: "xgmi"}

    def p2p_link_info(self, which: int) -> dict:
        """The link to the attached lower (which=0) / upper (1) neighbour's GPU (fdtd_p2p_link_info)."""
        a = np.full(8, -1, np.int32)
        self._ck(self.lib.fdtd_p2p_link_info(self._ctx, int(which), _ptr(a)), "p2p_link_info")
        return {"device": int(a[0]), "mapping": {1: "same-process", 2: "ipc"}.get(int(a[1]), None),
                "link": self.LINK_TYPES.get(int(a[2]), None if a[2] < 0 else f"type{int(a[2])}"), "hops": int(a[3]),
                "perf_rank": int(a[4]), "access": int(a[5]), "native_atomics": int(a[6]), "same_device": bool(a[7] == 1)}

    def schedule_info(self) -> dict:
        """Launches per timestep, lag, tiling and halo transport of this context (fdtd_schedule_info)."""
        a = np.zeros(8, np.int32)
        self._ck(self.lib.fdtd_schedule_info(self._ctx, _ptr(a)), "schedule_info")
        return {"launches_per_timestep": int(a[0]), "lag_planes": int(a[1]), "resident": bool(a[1] == -1), "rows_per_strip": int(a[2]),
                "blocks_per_sweep": int(a[3]),
                "transport": ("none", "p2p", "rccl", "linked", "external")[int(a[4])] if 0 <= a[4] <= 4 else None,
                "xcd_shares_weighted": bool(a[5]), "xcd_adaptations": int(a[6]), "timesteps_per_launch_max": int(a[7])}

    def comm_nranks(self) -> int:
        """Ranks of the RCCL communicator attached to this context (0: none)."""
        n = C.c_int(0)
        self._ck(self.lib.fdtd_comm_nranks(self._ctx, C.byref(n)), "comm_nranks")
        return n.value

    def comm_init(self, uid: bytes):
        if len(uid) != 128:
            raise ValueError("unique id must be 128 bytes")
        buf = C.create_string_buffer(uid, 128)
        self._ck(self.lib.fdtd_comm_init(self._ctx, buf), "comm_init")

    # -- fields -------------------------------------------------------------------------------
    def get_field(self, kind: int, comp: int) -> np.ndarray:
        out = np.empty(self.local_shape, np.float32)
        self._ck(self.lib.fdtd_get_field(self._ctx, int(kind), int(comp), _ptr(out)), "get_field")
        return out

    def set_field(self, kind: int, comp: int, a: np.ndarray):
        a = _arr(a, np.float32)
        if a.shape != self.local_shape:
            raise ValueError("field shape mismatch")
        self._ck(self.lib.fdtd_set_field(self._ctx, int(kind), int(comp), _ptr(a)), "set_field")

    def fields(self):
        """All six components as [2][3][nk][ny][nx]."""
        return np.stack([np.stack([self.get_field(kind, c) for c in range(3)]) for kind in (KIND_V, KIND_I)])


def link(lower, upper):
    """fdtd_link: adjacent slabs of one process (or one FDTD_FLAG_LOOPBACK slab with itself)."""
    lower._ck(lower.lib.fdtd_link(lower._ctx, upper._ctx), "link")


def run_linked(engines, nsteps: int):
    """Step adjacent slabs that live in this process together (fdtd_link + fdtd_run_linked)."""
    lib = engines[0].lib
    for lo, hi in zip(engines[:-1], engines[1:]):
        lo._ck(lib.fdtd_link(lo._ctx, hi._ctx), "link")
    arr = (C.c_void_p * len(engines))(*[e._ctx for e in engines])
    rc = lib.fdtd_run_linked(arr, len(engines), int(nsteps))
    if rc != 0:   # the message sits in the context that failed, not necessarily the first one
        msgs = [f"rank {r}: {m.decode()}" for r, e in enumerate(engines) if (m := lib.fdtd_last_error(e._ctx))]
        raise FdtdError(f"run_linked failed ({rc}): " + "; ".join(msgs))


def comm_unique_id(lib: C.CDLL) -> bytes:
    buf = C.create_string_buffer(128)
    rc = lib.fdtd_comm_unique_id(buf)
    if rc != 0:
        msg = lib.fdtd_last_error(None)
        raise FdtdError(f"fdtd_comm_unique_id failed ({rc}): {msg.decode() if msg else ''}")
    return buf.raw


def farfield(lib: C.CDLL, pos, Js, Ms, k_wave: float, theta, phi, device: int = 0):
    """theta/phi: paired 1-D direction lists [rad].  Returns complex (E_theta, E_phi) * r."""
    pos = _arr(pos, np.float64)
    npts = pos.shape[0]
    Js = _arr(np.stack([np.real(Js), np.imag(Js)], -1), np.float64)
    Ms = _arr(np.stack([np.real(Ms), np.imag(Ms)], -1), np.float64)
    th, ph = _arr(theta, np.float64), _arr(phi, np.float64)
    if pos.shape != (npts, 3) or Js.shape != (npts, 3, 2) or Ms.shape != (npts, 3, 2) or th.shape != ph.shape:
        raise ValueError("farfield argument shapes")
    eth = np.zeros((th.size, 2), np.float64)
    eph = np.zeros((th.size, 2), np.float64)
    rc = lib.fdtd_farfield(int(device), int(npts), _ptr(pos), _ptr(Js), _ptr(Ms), float(k_wave), int(th.size),
                           _ptr(th), _ptr(ph), _ptr(eth), _ptr(eph))
    if rc != 0:
        msg = lib.fdtd_last_error(None)
        raise FdtdError(f"fdtd_farfield failed ({rc}): {msg.decode() if msg else ''}")
    return eth[:, 0] + 1j * eth[:, 1], eph[:, 0] + 1j * eph[:, 1]

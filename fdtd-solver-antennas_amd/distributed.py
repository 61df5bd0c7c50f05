"""z-slab multi-GPU plumbing: one process per GPU, torch.distributed for bootstrap and reductions.

The reference is single-process (no collective call sites); the decomposition is new.
Two halo transports drive the same C ABI:

  * "rccl"  (product, GPU): the 128-byte RCCL unique id is broadcast through torch.distributed and
    handed to fdtd_comm_init(); the per-half-step ncclSend/ncclRecv of one Ix,Iy (up) / Vx,Vy (down)
    plane then lives inside libfdtd_hip.so's step loop on a second HIP stream, overlapped with the
    interior update (csrc/api.hip: step_loop/exchange).
  * "host"  (any engine exporting the ABI; used by the gloo CPU tests and as a debugging fallback):
    fdtd_half_step + fdtd_halo_get/put, planes shipped with torch.distributed send/recv.

Energy end-criterion terms, port series and NF2FF surface accumulators are summed over ranks with
all_reduce (each rank only accumulates what its slab owns).
"""
from __future__ import annotations

from typing import Optional
import numpy as np

from . import _capi


class SlabComm:
    def __init__(self, transport: str = "auto"):
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed must be initialised (one process per GPU)")
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.backend = dist.get_backend()
        self.transport = transport
        self.sim = None
        self.transport_used = None
        self.rccl_error = None

    # -- tensors on the right device for the process group ------------------------------------------
    def _to_t(self, a: np.ndarray):
        import torch
        t = torch.from_numpy(np.ascontiguousarray(a))
        return t.cuda() if self.backend == "nccl" else t

    def allreduce(self, a: np.ndarray) -> np.ndarray:
        a = np.asarray(a)
        if np.iscomplexobj(a):
            return self.allreduce(a.real) + 1j * self.allreduce(a.imag)
        t = self._to_t(a.astype(np.float64))
        self.dist.all_reduce(t)
        return t.cpu().numpy().reshape(a.shape)

    def barrier(self):
        self.dist.barrier()

    # -- transport selection ---------------------------------------------------------------------------
    def attach(self, sim):
        self.sim = sim
        eng = sim.engine
        use_rccl = self.transport == "rccl" or (self.transport == "auto" and eng.backend.startswith("hip"))
        if self.world == 1:
            sim.external_transport = None
            return
        if use_rccl:
            uid = [_capi.comm_unique_id(eng.lib) if self.rank == 0 else None]
            self.dist.broadcast_object_list(uid, src=0)
            ok = 1.0
            try:
                eng.comm_init(uid[0])
            except _capi.FdtdError as exc:      # e.g. an RCCL set-up problem on this node
                ok, self.rccl_error = 0.0, str(exc)
            # every rank must take the same path: fall back to the host transport together
            if float(self.allreduce(np.array([ok]))[0]) == self.world:
                sim.external_transport = None
                self.transport_used = "rccl"
                return
        sim.external_transport = self
        self.transport_used = "host"

    # -- host transport: one exchange after each half-step ------------------------------------------
    def exchange(self, eng, which: int):
        """HALO_E_DOWN: bottom Vx,Vy plane -> rank-1 (ghost above there); HALO_H_UP: top Ix,Iy -> rank+1."""
        r, w = self.rank, self.world
        dst, src = (r - 1, r + 1) if which == _capi.HALO_E_DOWN else (r + 1, r - 1)
        ops, recv = [], None
        if 0 <= dst < w:
            ops.append(self.dist.P2POp(self.dist.isend, self._to_t(eng.halo_get(which)), dst))
        if 0 <= src < w:
            recv = self._to_t(np.empty((2, eng.ny, eng.nx), np.float32))
            ops.append(self.dist.P2POp(self.dist.irecv, recv, src))
        if ops:
            for req in self.dist.batch_isend_irecv(ops):
                req.wait()
        if recv is not None:
            eng.halo_put(which, recv.cpu().numpy())

    def run_steps(self, eng, nsteps: int):
        for _ in range(nsteps):
            eng.half_step(_capi.PHASE_E)
            self.exchange(eng, _capi.HALO_E_DOWN)
            eng.half_step(_capi.PHASE_H)
            self.exchange(eng, _capi.HALO_H_UP)

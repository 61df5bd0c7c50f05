"""z-slab multi-GPU plumbing: one process per GPU, torch.distributed for bootstrap and reductions.

The reference is single-process (no collective call sites); the decomposition is new.
Three halo transports drive the same C ABI:

  * "p2p"   (product default, GPU): every rank exports the IPC handle of its halo mailbox, the handles travel through
    torch.distributed, neighbours attach them (fdtd_p2p_attach) and from then on the update kernels themselves push
    the halo planes into the neighbour's mailbox and wait on its step-counter flags: two launches per timestep on one
    stream, no RCCL call and no event in the step loop (an 8-plane north-star slab steps in 32 us instead of 85 us).
  * "rccl"  (GPU): the 128-byte RCCL unique id is broadcast through torch.distributed and
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
    def __init__(self, transport: str = "auto", skip=()):
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed must be initialised (one process per GPU)")
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.backend = dist.get_backend()
        self.transport = transport
        self.skip = set(skip)        # "auto" only: transports not to try (a caller that has seen one fail at run time)
        self.sim = None
        self.transport_used = None
        self.rccl_error = None
        self.p2p_error = None
        self.ms_exchange = 0.0       # host transport: wall time spent in halo exchanges
        self.ms_selftest = 0.0       # p2p transport: wall time of the data-path self-test over the neighbour links

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

    def allreduce_max(self, a: np.ndarray) -> np.ndarray:
        t = self._to_t(np.asarray(a, np.float64))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return t.cpu().numpy().reshape(np.shape(a))

    def barrier(self):
        """Also the tear-down rule of the p2p transport: every rank must have finished its last run before any rank
        frees its engine (a neighbour's last H half-step still writes into this rank's mailbox)."""
        self.dist.barrier()

    # -- transport selection ---------------------------------------------------------------------------
    def _all_agree(self, ok: bool) -> bool:
        return float(self.allreduce(np.array([1.0 if ok else 0.0]))[0]) == self.world

    def attach(self, sim):
        """Pick the halo transport for sim.engine — every rank takes the same one:
        "p2p"  mailbox transport (kernels push halos into the neighbour's mailbox; IPC handles exchanged here),
        "rccl" grouped ncclSend/ncclRecv on a second stream inside the library,
        "host" fdtd_half_step + torch.distributed send/recv (any engine, used by the gloo CPU tests).
        "auto" = p2p, then rccl, then host, falling back TOGETHER when a rank cannot set one up."""
        self.sim = sim
        sim.comm = self              # Simulation.run: which transport, and the ladder when it fails in the first timesteps
        eng = sim.engine
        if self.world == 1:
            sim.external_transport = None
            return
        on_gpu = eng.backend.startswith("hip")
        want = self.transport
        if want == "p2p" or (want == "auto" and "p2p" not in self.skip):
            eligible = on_gpu and not sim.mur_enable.any() and eng.nk >= 2
            if self._all_agree(eligible):
                ok, blobs = True, [None] * self.world
                try:
                    mine = eng.p2p_export()
                except _capi.FdtdError as exc:
                    ok, mine, self.p2p_error = False, b"", str(exc)
                self.dist.all_gather_object(blobs, mine)
                if self._all_agree(ok):
                    try:
                        eng.p2p_attach(blobs[self.rank - 1] if self.rank > 0 else None,
                                       blobs[self.rank + 1] if self.rank + 1 < self.world else None)
                    except _capi.FdtdError as exc:
                        ok, self.p2p_error = False, str(exc)
                    if self._all_agree(ok):
                        self.dist.barrier()
                        import time
                        t0 = time.perf_counter()
                        try:                       # tokens across every neighbour link before a timestep depends on them
                            eng.p2p_selftest(0x5E1F0001)
                        except _capi.FdtdError as exc:
                            ok, self.p2p_error = False, str(exc)
                        self.ms_selftest = (time.perf_counter() - t0) * 1e3
                    if self._all_agree(ok):
                        sim.external_transport = None
                        self.transport_used = "p2p"
                        return
                    eng.p2p_detach()
                    self.dist.barrier()
                if self.p2p_error:
                    import sys
                    print(f"[fdtd-hip rank {self.rank}] p2p halo transport not usable: {self.p2p_error}", file=sys.stderr, flush=True)
            if want == "p2p":
                raise RuntimeError(f"p2p halo transport unavailable: {self.p2p_error or 'not eligible (Mur faces, CPU engine or 1-plane slab)'}")
        if (want == "rccl" or (want == "auto" and "rccl" not in self.skip)) and on_gpu:
            uid = [_capi.comm_unique_id(eng.lib) if self.rank == 0 else None]
            self.dist.broadcast_object_list(uid, src=0)
            ok = True
            try:
                eng.comm_init(uid[0])
            except _capi.FdtdError as exc:      # e.g. an RCCL set-up problem on this node
                ok, self.rccl_error = False, str(exc)
            if self._all_agree(ok):
                sim.external_transport = None
                self.transport_used = "rccl"
                return
        sim.external_transport = self
        self.transport_used = "host"

    # -- host transport: one exchange after each half-step ------------------------------------------
    def exchange(self, eng, which: int):
        """HALO_E_DOWN: bottom Vx,Vy plane -> rank-1 (ghost above there); HALO_H_UP: top Ix,Iy -> rank+1."""
        import time
        t0 = time.perf_counter()
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
        self.ms_exchange += (time.perf_counter() - t0) * 1e3

    def run_steps(self, eng, nsteps: int):
        if nsteps > 0 and int(eng.step) == 0:      # the H halo of "step -1": the initial fields (non-zero initial fields decompose too)
            self.exchange(eng, _capi.HALO_H_UP)
        for _ in range(nsteps):
            eng.half_step(_capi.PHASE_E)
            self.exchange(eng, _capi.HALO_E_DOWN)
            eng.half_step(_capi.PHASE_H)
            self.exchange(eng, _capi.HALO_H_UP)

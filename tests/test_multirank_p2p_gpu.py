"""One process per rank on the GPU: z-slabs coupled by the P2P mailbox transport, mailboxes shared through HIP IPC
handles that travel over torch.distributed (gloo here).  All ranks sit on device 0 of the one-GPU test box; on a
multi-GPU node the same code maps the neighbours' mailboxes across xGMI.  The rank-concatenated fields must equal
the single-slab run bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, pkg
from helpers import patch_sim

pytestmark = pytest.mark.gpu
SHAPE = (48, 44, 36)
STEPS = 150


def _worker(rank, world, port, out_dir, one_launch, fault=False):
    import torch.distributed as dist
    if one_launch:      # every rank steps with ONE launch per timestep, the mailbox protocol inside it (read at fdtd_create)
        os.environ["FDTD_WAVEFRONT"] = "1"
    if fault:           # test hook: the halo waits of the run() call that covers timestep 5 expect tags nobody sends (20 us bound)
        os.environ["FDTD_P2P_FAULT_STEP"] = "5"
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    capi = pkg("_capi")
    lib = capi.load_hip_library()
    s = patch_sim(*SHAPE, cpml_cells=8, nr_ts=STEPS)
    e = s.build(lib, rank=rank, world=world, device=0)
    comm = pkg("distributed").SlabComm(transport="auto" if fault else "p2p")
    comm.attach(s)
    assert comm.transport_used == "p2p" and s.external_transport is None
    if world == 2:      # the host-level driver: chunks of 40 steps with the all-reduced energy criterion in between
        st = s.run(check_every=40, allreduce=comm.allreduce, log=lambda *_: None)
        assert st.steps == STEPS
        # the fault: every rank saw its halo waits time out in the first chunk, all of them rebuilt their slab and went down the ladder
        # together (RCCL refuses two ranks on one device -> host copies on this box); the run is the valid run from the initial state
        assert st.transports_failed == (("p2p",) if fault else ()) and comm.transport_used == ("host" if fault else "p2p")
        e = s.engine
    else:
        for n in (1, 60, STEPS - 61):
            e.run(n)
    u, i = s.port_series(comm.allreduce)[0]
    boxes = s.nf2ff_boxes(comm.allreduce)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), fields=e.fields(), u=u, i=i, k0=e.k0, nk=e.nk, step=e.step,
             **{f"box{n}": b for n, b in enumerate(boxes)})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,one_launch,fault", [(2, False, False), (3, False, False), (3, True, False), (2, False, True)])
def test_p2p_ranks_in_separate_processes_equal_one_slab(hip_lib, tmp_path, world, one_launch, fault):
    """... and (fault) `Simulation.run` of a decomposed run whose halo transport fails in the first timesteps: every rank takes the next
    transport, the result is still the single-slab result."""
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_worker, args=(world, port, str(tmp_path), one_launch, fault), nprocs=world, join=True)
    s = patch_sim(*SHAPE, cpml_cells=8, nr_ts=STEPS)
    e = s.build(hip_lib)
    e.run(STEPS)
    r = [np.load(tmp_path / f"rank{q}.npz") for q in range(world)]
    assert all(int(x["step"]) == STEPS for x in r) and sum(int(x["nk"]) for x in r) == SHAPE[2]
    both = np.concatenate([x["fields"] for x in r], axis=2)
    ref = e.fields()
    assert np.abs(ref).max() > 0
    assert np.array_equal(both.view(np.uint32), ref.view(np.uint32))
    u, i = s.port_series()[0]
    assert np.allclose(r[0]["u"], u, rtol=1e-12, atol=0) and np.allclose(r[-1]["i"], i, rtol=1e-12, atol=1e-300)
    for n, b in enumerate(s.nf2ff_boxes()):      # rank-summed running-DFT surfaces == single-slab surfaces
        assert np.allclose(r[0][f"box{n}"], b, rtol=1e-12, atol=1e-30) and np.allclose(r[-1][f"box{n}"], b, rtol=1e-12, atol=1e-30)

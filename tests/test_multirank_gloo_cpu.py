"""N > 1 on CPU: two processes, gloo, z-slab decomposition with the host halo transport, the oracle
standing in as the per-rank engine.  The two-slab run must reproduce the single-slab run bit for bit
(fields), and the rank-summed port series / NF2FF surfaces / energy must match."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, pkg
from helpers import patch_sim

SHAPE = (40, 38, 36)
STEPS = 120


def _worker(rank, world, port, out_dir):
    import ctypes
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = pkg("_capi").bind(ctypes.CDLL(os.path.join(ROOT, "oracle", "libfdtd_oracle.so")))
    s = patch_sim(*SHAPE, cpml_cells=8, nr_ts=STEPS)
    e = s.build(lib, rank=rank, world=world)
    comm = pkg("distributed").SlabComm(transport="auto")     # CPU engine: p2p and rccl are not eligible -> all ranks agree on "host"
    comm.attach(s)
    assert s.external_transport is comm and comm.transport_used == "host"
    st = s.run(check_every=40, allreduce=comm.allreduce)
    u, i = s.port_series(comm.allreduce)[0]
    boxes = s.nf2ff_boxes(comm.allreduce)
    sv, si = e.energy()
    en = comm.allreduce(np.array([sv, si]))
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), fields=e.fields(), u=u, i=i, k0=e.k0, nk=e.nk, en=en,
             steps=st.steps, **{f"box{n}": b for n, b in enumerate(boxes)})
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_equal_one(oracle_lib, tmp_path, world):
    """world 3: the cost-weighted partition (simulation.slab_partition) gives the two end ranks, which own all z-CPML
    planes, fewer planes than the middle one — the uneven slabs must still reproduce the single slab bit for bit."""
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    s = patch_sim(*SHAPE, cpml_cells=8, nr_ts=STEPS)
    e = s.build(oracle_lib)
    e.run(STEPS)
    r = [np.load(tmp_path / f"rank{q}.npz") for q in range(world)]
    nks = [int(q["nk"]) for q in r]
    assert int(r[0]["k0"]) == 0 and sum(nks) == SHAPE[2]
    assert [(int(q["k0"]), int(q["nk"])) for q in r] == s.slabs(world)
    if world == 3:
        assert nks[1] > nks[0] and nks[1] > nks[2]
    assert int(r[0]["steps"]) == STEPS
    both = np.concatenate([q["fields"] for q in r], axis=2)
    ref = e.fields()
    assert np.abs(ref).max() > 0
    assert np.array_equal(both.view(np.uint32), ref.view(np.uint32))
    u, i = s.port_series()[0]
    assert np.allclose(r[0]["u"], u, rtol=1e-12, atol=0) and np.allclose(r[-1]["i"], i, rtol=1e-12, atol=1e-300)
    for n, b in enumerate(s.nf2ff_boxes()):
        for q in r:
            assert np.allclose(q[f"box{n}"], b, rtol=1e-12, atol=1e-30)
    sv, si = e.energy()
    assert np.allclose(r[0]["en"], [sv, si], rtol=1e-12)


def _worker_api(rank, world, port, out_dir):
    """The plugin-level path: openems_api.openEMS(rank=, world=, comm=) — Run() on every rank, post-processing on rank 0 ALONE."""
    import ctypes
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import tutorial_scene
    lib = pkg("_capi").bind(ctypes.CDLL(os.path.join(ROOT, "oracle", "libfdtd_oracle.so")))
    comm = pkg("distributed").SlabComm(transport="auto")
    r = tutorial_scene.build_and_run(lib, os.path.join(out_dir, f"run{rank}"), nr_ts=400, end_criteria=0, f_ff=2e9, post=(rank == 0),
                                     rank=rank, world=world, comm=comm)
    if rank == 0:
        np.savez(os.path.join(out_dir, "api_rank0.npz"), s11=r["s11"], Dmax=r["Dmax"], E=r["E_norm"], u=r["u"])
    dist.barrier()
    dist.destroy_process_group()


def test_plugin_level_two_ranks_postprocess_on_rank_zero_alone(oracle_lib, tmp_path):
    """in record mode the cross-rank reduction of the NF2FF faces had moved into CalcNF2FF, which made it a
    collective — a caller that post-processes on rank 0 only hung.  Run() now reduces the faces for the excitation's centre
    frequency where every rank is; CalcNF2FF(f0) and CalcPort on rank 0 alone return, and equal the single-rank run."""
    import torch.multiprocessing as mp
    import tutorial_scene
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_worker_api, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    two = np.load(tmp_path / "api_rank0.npz")
    one = tutorial_scene.build_and_run(oracle_lib, str(tmp_path / "single"), nr_ts=400, end_criteria=0, f_ff=2e9)
    assert one["nf2ff_mode"] == "record"
    assert np.allclose(two["u"], one["u"], rtol=1e-12, atol=0) and np.allclose(two["s11"], one["s11"], rtol=1e-9)
    assert abs(float(two["Dmax"]) - one["Dmax"]) < 1e-9 * one["Dmax"] and np.allclose(two["E"], one["E_norm"], rtol=1e-9)

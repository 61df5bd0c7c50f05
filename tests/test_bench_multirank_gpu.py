"""bench.py as the driver launches it (python -m torch.distributed.run, one rank per GPU), kept alive on the one-GPU test
box: all ranks share device 0 (FDTD_BENCH_FORCE_DEVICE) and bootstrap over gloo, because RCCL refuses several ranks on
one device — everything else is the product path: z-slab decomposition of ONE grid, halo transport inside
libfdtd_hip.so, the JSON line with the `multi_gpu` block that lets a future 8-GPU record show whether the ranks really
coupled.  The decomposed runs reproduce the single-rank port series (port_u_l2) whatever the transport."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
ARGS = ["--steps", "2", "--warmup", "1", "--ts-per-step", "100", "--workload", "C2", "--no-cpu-baseline", "--no-hbm-point",
        "--prefill-seconds", "0"]


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _bench(nranks, extra, env_extra=None, launcher=True):
    env = dict(os.environ)
    env.update({"FDTD_BENCH_FORCE_DEVICE": "0", "FDTD_BENCH_DIST_BACKEND": "gloo", "MASTER_ADDR": "127.0.0.1"})
    env.update(env_extra or {})
    cmd = [sys.executable]
    if launcher:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nranks), "--master-addr", "127.0.0.1",
                "--master-port", str(_free_port())]
    cmd += [os.path.join(ROOT, "bench.py"), "--gpus", str(nranks)] + ARGS + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.fixture(scope="module")
def single():
    out = _bench(1, [], launcher=False)
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["config"]["fields_finite"]
    assert out["roofline"]["frac"] > 0 and out["roofline"]["resident"] == "infinity-cache"
    assert 0.5 < out["roofline"]["kernels_over_timestep"] <= 1.03
    return out


@pytest.mark.parametrize("nranks,halo", [(2, "p2p"), (2, "auto"), (3, "host")])
def test_bench_under_torchrun_couples_the_ranks(single, nranks, halo):
    out = _bench(nranks, ["--halo", halo])
    assert out["n_gpus"] == nranks and out["value"] > 0 and out["scaling"] == "strong"
    assert out["config"]["fields_finite"] and out["config"]["timesteps_total"] == single["config"]["timesteps_total"]
    mg = out["multi_gpu"]
    assert mg["ranks_seen"] == nranks and len(mg["per_rank"]) == nranks
    assert mg["transport_used"] == ("p2p" if halo in ("p2p", "auto") else "host") and mg["transports_failed"] == []
    assert f"z-slab x{nranks}" in out["config"]["parallelism"] and mg["transport_used"] in out["config"]["parallelism"]
    assert sum(r["slab_planes"] for r in mg["per_rank"]) == 40
    # the record describes itself: partition, per-rank planes / z-layer planes / time / schedule, and (p2p) what the runtime knows
    # about the way to each neighbour's GPU — here every rank sits on device 0
    assert mg["partition"] == "cost" and mg["z_layer_plane_cost"] == 1.4 and mg["rank_time_max_over_mean"] >= 1.0
    assert sum(r["slab_z_cpml_planes"] for r in mg["per_rank"]) == 21 and [r["slab_k0"] for r in mg["per_rank"]][0] == 0
    assert all(r["launches_per_timestep"] in (0, 1, 2) and r["us_per_timestep"] > 0 for r in mg["per_rank"])
    if mg["transport_used"] == "p2p":
        for r in mg["per_rank"]:
            assert r["launches_per_timestep"] == 2 and r["ms_p2p_selftest"] > 0
            links = [l for l in (r["link_down"], r["link_up"]) if l is not None]
            assert len(links) == (2 if 0 < r["rank"] < nranks - 1 else 1)
            assert all(l["same_device"] and l["device"] == 0 and l["mapping"] == "ipc" for l in links)
        if nranks == 3:
            assert [r["slab_planes"] for r in mg["per_rank"]][1] > mg["per_rank"][0]["slab_planes"]
    assert mg["all_slabs_excited"]                      # slabs without the port are non-zero only through their halos
    assert mg["rccl_nranks"] == 0                       # no RCCL communicator on the shared device
    if halo == "host":
        assert all(r["ms_halo_host_exchange"] > 0 for r in mg["per_rank"])
    # the decomposed run IS the single-slab run: same port voltage series (float64 sums in another order)
    assert abs(out["config"]["port_u_l2"] - single["config"]["port_u_l2"]) <= 1e-9 * single["config"]["port_u_l2"]
    assert single["config"]["port_u_l2"] > 0


def test_bench_takes_the_next_transport_when_one_fails_at_run_time(single):
    """The P2P transport sets up and passes its self-test, then its halo waits time out once timesteps depend on them (test hook:
    the waits of a run covering $FDTD_P2P_FAULT_STEP expect tags nobody sends; 20 us bound).  Every rank sees the error, all of them
    rebuild their slab and go down the ladder together — RCCL refuses two ranks on one device, so this box ends on the host
    transport — and the line is the valid result of the SAME run: same timestep count, same port series, the failure on record."""
    out = _bench(2, ["--halo", "auto"], env_extra={"FDTD_P2P_FAULT_STEP": "10"})
    mg = out["multi_gpu"]
    assert [f["transport"] for f in mg["transports_failed"]] == ["p2p"] and "halo wait timed out" in mg["transports_failed"][0]["error_on_this_rank"]
    assert mg["transport_used"] == "host" and mg["ranks_seen"] == 2 and mg["all_slabs_excited"]
    assert out["config"]["fields_finite"] and out["config"]["timesteps_total"] == single["config"]["timesteps_total"]
    assert abs(out["config"]["port_u_l2"] - single["config"]["port_u_l2"]) <= 1e-9 * single["config"]["port_u_l2"]
    # ... and an explicitly requested transport is not replaced behind the caller's back
    env = dict(os.environ)
    env.update({"FDTD_BENCH_FORCE_DEVICE": "0", "FDTD_BENCH_DIST_BACKEND": "gloo", "MASTER_ADDR": "127.0.0.1", "FDTD_P2P_FAULT_STEP": "10"})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2"] + ARGS + ["--halo", "p2p"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "failed at run time" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_default_nccl_bootstrap_at_world_one():
    """The `nccl` (= RCCL) process-group branch bench.py takes by default under torch.distributed.run, at world 1."""
    out = _bench(1, [], env_extra={"FDTD_BENCH_DIST_BACKEND": "nccl"})
    assert out["n_gpus"] == 1 and out["value"] > 0 and "multi_gpu" not in out

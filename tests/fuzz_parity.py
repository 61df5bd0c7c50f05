#!/usr/bin/env python3
"""Randomised differential run: libfdtd_hip.so against the CPU oracle on grids, boundaries and schedules nobody picked by hand.

Every case draws a grid shape (odd sizes, rows of 2 ... 65 four-cell groups, 12 ... 48 planes), a boundary per face (PEC / MUR / CPML, or
CPML on all six), a layer thickness, the operator form (class bytes / raw arrays), the kernel schedule (AUTO, two launches per timestep,
one launch per timestep forced, resident in registers forced), the tiling ($FDTD_TYS), the timesteps per launch ($FDTD_WF_MULTI), whether the XCD shares are measured
($FDTD_XCD_ADAPT), NF2FF faces as running sums / recorded samples / none, random initial fields, and a list of run() calls of random
length (the launches of several timesteps are cut at calls and at NF2FF sample steps).  The same list goes to both engines through the
same C ABI; compared: all six field components as IEEE values (bit for bit wherever the oracle's value is non-zero), the port series and
the energy to 1e-12, the NF2FF face spectra to 1e-6 of their largest entry (float32 sums in another order).

    python tests/fuzz_parity.py [--cases 60] [--seed 1] [--only N]     # on a GPU box; prints one line per case, exits 1 on a mismatch
    python tests/fuzz_parity.py --slabs ...     # decomposed runs: 2 ... 6 z-slabs of drawn partition and schedules, coupled through their P2P
                                                # mailboxes in one process (fdtd_run_linked), against ONE slab on the oracle ("lag" column: planes per slab)

Lives under tests/ because it loads the oracle (test infrastructure); tests/test_round3_gpu.py::test_randomised_cases_equal_the_oracle
runs a fixed-seed batch of it in the -m gpu suite.
"""
import argparse
import ctypes
import importlib
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "fdtd-solver-antennas_amd"
ENV_KNOBS = ("FDTD_TYS", "FDTD_WF_MULTI", "FDTD_XCD_ADAPT", "FDTD_WF_LAG", "FDTD_RESIDENT", "FDTD_RES_CHUNK", "FDTD_MUR_APPLY_PASS")


def _mod(name):
    return importlib.import_module(PKG + "." + name)


def draw_case(rng, mur=False):
    """One case as a plain dict (printable, reproducible from --seed / --only)."""
    nx = int(rng.choice([rng.integers(8, 40), rng.integers(40, 141), 4 * rng.integers(3, 30) + 1, 4 * rng.integers(3, 30)]))
    ny = int(rng.choice([rng.integers(8, 30), rng.integers(30, 121)]))
    nz = int(rng.integers(12, 49))
    r = rng.random()
    if r < 0.4:
        kinds = ["CPML"] * 6
    elif r < 0.8:
        kinds = [str(rng.choice(["PEC", "CPML", "CPML"])) for _ in range(6)]
    else:
        kinds = [str(rng.choice(["PEC", "MUR", "CPML"])) for _ in range(6)]
    if rng.random() < 0.12:      # now and then a grid of a few thousand blocks per half-step (several blocks per CU, every XCD share long)
        nx, ny, nz = int(rng.integers(150, 260)), int(rng.integers(120, 220)), int(rng.integers(24, 44))
    if mur:      # --mur: every case has Mur faces (all six, or mixed with PEC / CPML) and mostly runs on the launch-per-half-step schedules
        kinds = ["MUR"] * 6 if rng.random() < 0.5 else [str(rng.choice(["PEC", "MUR", "MUR", "CPML"])) for _ in range(6)]
        if "MUR" not in kinds:
            kinds[int(rng.integers(0, 6))] = "MUR"
    cells_max = max(2, min(12, (min(nx, ny, nz) - 8) // 2))
    cells = int(rng.integers(2, cells_max + 1))
    has_mur = "MUR" in kinds
    sched = str(rng.choice(["auto", "direct", "wavefront", "resident"] if has_mur else ["auto", "direct", "wavefront", "wavefront", "resident"]))
    env = {}
    # round 4: AUTO steps small grids resident in registers; half of the AUTO cases keep the schedules AUTO took before (they still serve
    # every grid that does not fit the chip), and the resident launches are cut short now and then
    if sched == "auto" and rng.random() < 0.5:
        env["FDTD_RESIDENT"] = "0"
    if mur:
        if sched != "resident" and rng.random() < 0.8:
            env["FDTD_RESIDENT"] = "0"
        if rng.random() < 0.25:      # the apply pass as a launch of its own (three launches per timestep) instead of inside update_H (two)
            env["FDTD_MUR_APPLY_PASS"] = "1"
    if rng.random() < 0.3:
        env["FDTD_RES_CHUNK"] = str(int(rng.choice([1, 2, 7, 33])))
    if rng.random() < 0.4:
        env["FDTD_TYS"] = str(int(rng.choice([1, 2, 3, 4, 5, 7, 9, 16, 40])))
    if rng.random() < 0.5:
        env["FDTD_WF_MULTI"] = str(int(rng.choice([1, 2, 3, 5, 64])))
    if rng.random() < 0.3:
        env["FDTD_XCD_ADAPT"] = "0"
    if sched == "wavefront" and rng.random() < 0.3:
        env["FDTD_WF_LAG"] = str(int(rng.choice([0, 1, 2, 3])))
    nf = str(rng.choice(["none", "dft", "record"]))
    ncalls = int(rng.integers(1, 6))
    calls = [int(rng.integers(1, 90)) for _ in range(ncalls)]
    return {"shape": (nx, ny, nz), "kinds": kinds, "cells": cells, "classes": bool(rng.random() < 0.7), "sched": sched, "env": env,
            "nf2ff": nf, "calls": calls, "seed": int(rng.integers(1, 1 << 30))}


def run_case(case, hip, oracle):
    capi, wl, sc, simm = _mod("_capi"), _mod("workloads"), _mod("scene"), _mod("simulation")
    nx, ny, nz = case["shape"]
    w = wl.patch_workload("fuzz", nx=nx, ny=ny, nz=nz)
    vox = sc.voxelize(w.scene, w.grid)
    flags = {"auto": 0, "direct": capi.FLAG_KERNEL_DIRECT, "wavefront": capi.FLAG_KERNEL_WAVEFRONT, "resident": capi.FLAG_KERNEL_RESIDENT}[case["sched"]]
    total = sum(case["calls"])
    saved = {k: os.environ.pop(k, None) for k in ENV_KNOBS}
    os.environ.update(case["env"])
    out = []
    try:
        for lib in (hip, oracle):
            sim = simm.Simulation(w.grid, vox, f0=w.f0, fc=w.fc, boundary=case["kinds"], cpml_cells=case["cells"], nr_ts=total + 8,
                                  nf2ff_freqs=None if case["nf2ff"] == "none" else [w.f0, 1.3 * w.f0], use_classes=case["classes"],
                                  nf2ff_mode="dft" if case["nf2ff"] == "none" else case["nf2ff"])
            eng = sim.build(lib, flags=flags if lib is hip else 0)
            rng = np.random.default_rng(case["seed"])
            for kind in (0, 1):
                for c in range(3):
                    eng.set_field(kind, c, (1e-3 * rng.standard_normal(eng.local_shape)).astype(np.float32))
            for n in case["calls"]:
                eng.run(n)
            out.append((sim, eng))
    finally:
        for k in ENV_KNOBS:
            os.environ.pop(k, None)
            if saved[k] is not None:
                os.environ[k] = saved[k]
    (sh, eh), (so, eo) = out
    problems = []
    fh, fo = eh.fields(), eo.fields()
    if not np.isfinite(fo).all() or not np.abs(fo).max() > 0:
        problems.append("oracle fields not finite / all zero")
    if not np.array_equal(fh, fo):
        bad = np.argwhere(fh != fo)
        problems.append(f"fields differ at {len(bad)} of {fo.size} entries, first {tuple(bad[0])}: {fh[tuple(bad[0])]!r} vs {fo[tuple(bad[0])]!r}")
    else:
        nzm = fo != 0
        if not np.array_equal(fh[nzm].view(np.uint32), fo[nzm].view(np.uint32)):
            problems.append("fields equal as values but not as bits where the oracle is non-zero")
    uh, uo = sh.port_series()[0], so.port_series()[0]
    for q, name in ((0, "port voltage"), (1, "port current")):
        a, b = np.asarray(uh[q], float), np.asarray(uo[q], float)
        if a.shape != b.shape or np.linalg.norm(a - b) > 1e-12 * max(np.linalg.norm(b), 1e-300):
            problems.append(f"{name} series differs")
    ea, eb = np.array(eh.energy()), np.array(eo.energy())
    if np.abs(ea - eb).max() > 1e-12 * np.abs(eb).max():
        problems.append(f"energy {ea!r} vs {eb!r}")
    if case["nf2ff"] != "none":
        bh, bo = sh.nf2ff_boxes(), so.nf2ff_boxes()
        worst = max(float(np.abs(np.asarray(a) - np.asarray(b)).max()) / max(float(np.abs(b).max()), 1e-300) for a, b in zip(bh, bo))
        if len(bh) != len(bo) or not worst <= 1e-6:
            problems.append(f"NF2FF face spectra differ ({worst:.2e} of the largest entry)")
    info = eh.schedule_info()
    return problems, info


def draw_slab_case(rng):
    """A decomposed run: 2 ... 6 z-slabs in this process on one GPU, coupled only through their P2P mailboxes (the transport of
    `bench.py --gpus N`), each slab under its own drawn schedule, against ONE slab on the oracle."""
    world = int(rng.integers(2, 7))
    nx = int(rng.choice([rng.integers(12, 60), rng.integers(60, 161), 4 * rng.integers(4, 30) + 1]))
    ny = int(rng.choice([rng.integers(10, 40), rng.integers(40, 141)]))
    nz = int(rng.integers(max(14, 3 * world + 2), 64))
    if rng.random() < 0.15:
        nx, ny = int(rng.integers(160, 300)), int(rng.integers(150, 260))       # slabs of more than one round of resident blocks
    kinds = ["CPML"] * 6 if rng.random() < 0.6 else [str(rng.choice(["PEC", "CPML", "CPML"])) for _ in range(6)]
    cells = int(rng.integers(2, max(2, min(12, (min(nx, ny, nz) - 8) // 2)) + 1))
    env = {}
    if rng.random() < 0.3:
        env["FDTD_TYS"] = str(int(rng.choice([1, 2, 3, 5, 7, 16])))
    sched = [str(rng.choice(["auto", "direct", "wavefront"])) for _ in range(world)]
    # All these slabs share ONE GPU here (AUTO then takes two launches: api.hip, wavefront_active).  Forcing one launch per timestep on some of
    # them keeps blocks resident that spin on a neighbour's halo; with thousands of small blocks per sweep (one-row strips of a 199 x 233
    # plane, six slabs) the spinning blocks of four such kernels held every slot of the chip and the two-launch kernels they waited for
    # never got one — a property of the shared GPU, not of the protocol.  Forced mixtures therefore stay below one round of resident blocks.
    if "wavefront" in sched and any(x != "wavefront" for x in sched):
        env.pop("FDTD_TYS", None)
        nx, ny = min(nx, 120), min(ny, 100)
        cells = min(cells, max(2, (min(nx, ny, nz) - 8) // 2))
    return {"world": world, "shape": (nx, ny, nz), "kinds": kinds, "cells": cells, "classes": bool(rng.random() < 0.7),
            "partition": str(rng.choice(["cost", "even"])), "sched": sched,
            "env": env, "nf2ff": str(rng.choice(["none", "dft"])), "calls": [int(rng.integers(1, 70)) for _ in range(int(rng.integers(1, 5)))],
            "seed": int(rng.integers(1, 1 << 30))}


def run_slab_case(case, hip, oracle):
    capi, wl, sc, simm = _mod("_capi"), _mod("workloads"), _mod("scene"), _mod("simulation")
    nx, ny, nz = case["shape"]
    world = case["world"]
    w = wl.patch_workload("fuzz", nx=nx, ny=ny, nz=nz)
    vox = sc.voxelize(w.scene, w.grid)
    fl = {"auto": 0, "direct": capi.FLAG_KERNEL_DIRECT, "wavefront": capi.FLAG_KERNEL_WAVEFRONT}
    total = sum(case["calls"])

    def make():
        return simm.Simulation(w.grid, vox, f0=w.f0, fc=w.fc, boundary=case["kinds"], cpml_cells=case["cells"], nr_ts=total + 8,
                               nf2ff_freqs=None if case["nf2ff"] == "none" else [w.f0], use_classes=case["classes"])
    saved = {k: os.environ.pop(k, None) for k in ENV_KNOBS}
    os.environ.update(case["env"])
    try:
        so = make()
        eo = so.build(oracle)
        sims = [make() for _ in range(world)]
        engs = [s.build(hip, rank=r, world=world, flags=fl[case["sched"][r]], partition=case["partition"]) for r, s in enumerate(sims)]
        blobs = [e.p2p_export() for e in engs]
        for r, e in enumerate(engs):
            e.p2p_attach(blobs[r - 1] if r > 0 else None, blobs[r + 1] if r + 1 < world else None)
        rng = np.random.default_rng(case["seed"])
        for kind in (0, 1):
            for c in range(3):
                g = (1e-3 * rng.standard_normal(eo.local_shape)).astype(np.float32)
                eo.set_field(kind, c, g)
                for e in engs:
                    e.set_field(kind, c, np.ascontiguousarray(g[e.k0:e.k0 + e.nk]))
        for n in case["calls"]:
            eo.run(n)
            capi.run_linked(engs, n)
    finally:
        for k in ENV_KNOBS:
            os.environ.pop(k, None)
            if saved[k] is not None:
                os.environ[k] = saved[k]
    problems = []
    fo, fh = eo.fields(), np.concatenate([e.fields() for e in engs], axis=2)
    if not np.isfinite(fo).all() or not np.abs(fo).max() > 0:
        problems.append("oracle fields not finite / all zero")
    if not np.array_equal(fh, fo):
        bad = np.argwhere(fh != fo)
        problems.append(f"fields differ at {len(bad)} of {fo.size} entries, first {tuple(bad[0])}: {fh[tuple(bad[0])]!r} vs {fo[tuple(bad[0])]!r}")
    uo = so.port_series()[0]
    for q, name in ((0, "port voltage"), (1, "port current")):
        a, b = sum(np.asarray(s.port_series()[0][q], float) for s in sims), np.asarray(uo[q], float)
        if a.shape != b.shape or np.linalg.norm(a - b) > 1e-12 * max(np.linalg.norm(b), 1e-300):
            problems.append(f"{name} series differs")
    if case["nf2ff"] != "none":
        for b, *parts in zip(so.nf2ff_boxes(), *[s.nf2ff_boxes() for s in sims]):
            if np.abs(sum(parts) - b).max() > 1e-6 * max(np.abs(b).max(), 1e-300):
                problems.append("NF2FF face spectra differ")
                break
    info = {"launches_per_timestep": "/".join(str(e.schedule_info()["launches_per_timestep"]) for e in engs),
            "lag_planes": "/".join(str(e.nk) for e in engs), "timesteps_per_launch_max": 1}
    return problems, info


def load_libs():
    capi = _mod("_capi")
    so = os.path.join(ROOT, "oracle", "libfdtd_oracle.so")
    src = os.path.join(ROOT, "oracle", "fdtd_oracle.c")
    if not os.path.isfile(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    import torch  # noqa: F401  (warms the ROCm runtime)
    return capi.load_hip_library(), capi.bind(ctypes.CDLL(so))


def run_batch(ncases, seed, hip, oracle, only=None, log=print, slabs=False, mur=False):
    rng = np.random.default_rng(seed)
    failed = []
    for n in range(ncases):
        case = draw_slab_case(rng) if slabs else draw_case(rng, mur)
        if only is not None and n != only:
            continue
        t0 = time.perf_counter()
        try:
            problems, info = (run_slab_case if slabs else run_case)(case, hip, oracle)
        except ValueError as exc:      # a drawn set-up the host layer refuses (e.g. layers that leave no room for the NF2FF box)
            log(f"case {n}: skipped ({exc}) {case}")
            continue
        except _mod("_capi").FdtdError as exc:      # an error from the library (a bounded wait that ran out, ...) is a failing case
            if "(-5): resident schedule" in str(exc) or "(-5): wavefront schedule" in str(exc):    # ... or a grid the schedule asked for by name cannot hold
                log(f"case {n}: refused ({str(exc)[:120]}) {case}")
                continue
            if "not starvation-free" in str(exc):    # ... except a drawn decomposition the library REFUSES: slabs sharing the test GPU that could pin every workgroup slot
                log(f"case {n}: refused ({str(exc)[:150]}...) {case}")
                continue
            log(f"case {n}: FAIL (library error) {case}  -> {exc}")
            failed.append((n, case, [str(exc)]))
            continue
        tag = "ok  " if not problems else "FAIL"
        log(f"case {n}: {tag} {time.perf_counter() - t0:5.1f} s  launches/ts {info['launches_per_timestep']} lag {info['lag_planes']} "
            f"ts/launch {info['timesteps_per_launch_max']}  {case}" + ("  -> " + "; ".join(problems) if problems else ""))
        if problems:
            failed.append((n, case, problems))
    return failed


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--only", type=int, default=None)
    ap.add_argument("--slabs", action="store_true", help="decomposed runs: 2 ... 6 P2P slabs in one process against one slab on the oracle")
    ap.add_argument("--mur", action="store_true", help="every case with Mur faces, mostly on the two / three launches per timestep")
    args = ap.parse_args()
    hip, oracle = load_libs()
    failed = run_batch(args.cases, args.seed, hip, oracle, args.only, log=lambda s: print(s, flush=True), slabs=args.slabs, mur=args.mur)
    print(f"{len(failed)} failing case(s) of {args.cases} (seed {args.seed})", flush=True)
    sys.exit(1 if failed else 0)


if __name__ == "__main__":
    main()

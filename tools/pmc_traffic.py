#!/usr/bin/env python3
"""HBM-side traffic per launch of the update kernels from rocprofv3 PMC passes, as MI355X_MICROARCH.md prescribes:
separate --pmc runs for FETCH_SIZE and WRITE_SIZE together with --kernel-trace only, KiB units, FETCH_SIZE doubled
(gfx950 correction).  Run ON the GPU box from the repo root:

    python3 tools/pmc_traffic.py NS gpurun_out/pmc_NS.json

The profiled program is `python3 bench.py` itself (short run), started directly after `--`.
"""
import csv, glob, json, os, subprocess, sys, collections

wl, out = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
raw = collections.defaultdict(dict)
timesteps_total = None
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    d = os.path.join(root, "gpurun_out", f"pmc_{wl}_{ctr}")
    subprocess.run(["rm", "-rf", d])
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "pmc", "--",
           "python3", os.path.join(root, "bench.py"), "--workload", wl, "--steps", "2", "--warmup", "1", "--ts-per-step", "10", "--no-cpu-baseline", "--no-hbm-point", "--no-small-grid-point", "--prefill-seconds", "0"]
    r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    if r.returncode != 0:
        sys.stderr.write(r.stderr[-2000:])
        raise SystemExit(f"rocprofv3 failed for {ctr}")
    for line in r.stdout.splitlines():
        if line.startswith("{"):
            timesteps_total = json.loads(line)["config"]["timesteps_total"]
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter csv under {d}")
    acc = collections.defaultdict(list)
    with open(files[0]) as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != ctr:
                continue
            name = row["Kernel_Name"]
            short = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
            acc[short].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        raw[k][ctr] = {"launches": len(v), "mean": sum(v) / len(v)}
per = {}


def holds_several_timesteps(k):   # k_step<COEF, PML, P2P, MULTI, MUR>: the fourth template argument
    if "k_step<" not in k:
        return False
    a = [x.strip() for x in k[k.index("<") + 1:k.rindex(">")].split(",")]
    return len(a) >= 4 and a[3] == "true"


# launches of k_step<.., MULTI = true, ..> hold several timesteps: their traffic per TIMESTEP = per launch / (timesteps they stepped / launches)
single = sum(v["FETCH_SIZE"]["launches"] for k, v in raw.items() if "k_step" in k and not holds_several_timesteps(k) and "FETCH_SIZE" in v)
for k, v in raw.items():
    if ("k_update" in k or "k_step" in k) and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        rd = v["FETCH_SIZE"]["mean"] * 1024.0 * 2.0      # KiB, x2 gfx950 correction
        wr = v["WRITE_SIZE"]["mean"] * 1024.0
        per[k] = {"read_bytes_corrected": rd, "write_bytes": wr, "total_bytes": rd + wr}
        if holds_several_timesteps(k) and timesteps_total:
            tsl = (timesteps_total - single) / v["FETCH_SIZE"]["launches"]
            per[k].update({"timesteps_per_launch": tsl, "total_bytes_per_timestep": (rd + wr) / tsl})
json.dump({"raw": raw, "per_launch_traffic": per, "workload": wl, "timesteps_total": timesteps_total,
           "note": "separate --pmc passes; FETCH_SIZE x2 (gfx950 correction); KiB units; tools/pmc_traffic.py"}, open(out, "w"), indent=1)
print(json.dumps(per, indent=1))

#!/usr/bin/env python3
"""Per-launch means of arbitrary rocprofv3 PMC counters for the update kernels of a short `python3 bench.py` run (one
--pmc pass per group, with --kernel-trace only).  Run ON the GPU box from the repo root:

    python3 tools/pmc_counters.py NS gpurun_out/pmc_sq_NS.json SQ_WAVE_CYCLES,SQ_WAIT_ANY,... [second group] ...
"""
import csv, glob, json, os, subprocess, sys, collections

wl, out, groups = sys.argv[1], sys.argv[2], sys.argv[3:]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
res = collections.defaultdict(dict)
for gi, grp in enumerate(groups):
    d = os.path.join(root, "gpurun_out", f"pmcg_{wl}_{gi}")
    subprocess.run(["rm", "-rf", d])
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", *grp.split(","), "--output-format", "csv", "-d", d, "-o", "pmc", "--",
           "python3", os.path.join(root, "bench.py"), "--workload", wl, "--steps", "2", "--warmup", "1", "--ts-per-step", "10",
           "--no-cpu-baseline", "--no-hbm-point", "--no-small-grid-point"]
    r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    if r.returncode != 0:
        sys.stderr.write(r.stderr[-3000:])
        raise SystemExit(f"rocprofv3 failed for group {grp}")
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter csv under {d}")
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(files[0]) as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"]
            short = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
            if "k_update" in short or "k_step" in short:
                acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, ctrs in acc.items():
        for c, v in ctrs.items():
            res[k][c] = sum(v) / len(v)
    subprocess.run(["rm", "-rf", d])
json.dump({"workload": wl, "per_launch_mean": res}, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))

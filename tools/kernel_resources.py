"""Register / scratch / occupancy table of the update kernels (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel.

    python tools/kernel_resources.py [regex]        # runs here (no GPU needed)
"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "fdtd-solver-antennas_amd", "csrc"), "resources"],
                     capture_output=True, text=True).stderr
pat = re.compile(sys.argv[1] if len(sys.argv) > 1 else "k_step|k_update")
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark: +(Function Name|TotalSGPRs|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        continue
    if m.group(1) == "Function Name":
        cur = {"name": subprocess.run(["c++filt", m.group(2)], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
    elif cur is not None:
        cur[m.group(1).split(" ")[0]] = m.group(2)
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::|void ", "", r["name"]).split("(")[0]
    if pat.search(name):
        print(f"{name:48s} SGPR {r.get('TotalSGPRs','?'):>4s}  VGPR {r.get('VGPRs','?'):>4s}  scratch {r.get('ScratchSize','?'):>4s}  waves/SIMD {r.get('Occupancy','?'):>2s}  LDS {r.get('LDS','?')}")

"""Gaps between consecutive kernel dispatches of a rocprofv3 --kernel-trace database: per kernel name the mean duration, and the mean
idle time between the end of one dispatch and the start of the next on the device (launch-bound loops show up here, not in --stats).

    python tools/rocprof_db_gaps.py <rocprofv3 output dir> [skip first N dispatches]
"""
import glob
import sqlite3
import sys
from collections import defaultdict

db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
con = sqlite3.connect(db)
tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
view = "kernels" if "kernels" in tabs else next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
cols = [r[1] for r in con.execute(f"pragma table_info({view})")]
name_col = "name" if "name" in cols else "kernel_name"
rows = list(con.execute(f"select {name_col}, start, end from {view} order by start"))[skip:]
dur, gap_after, n = defaultdict(float), defaultdict(float), defaultdict(int)
for (nm, s, e), nxt in zip(rows, rows[1:] + [None]):
    key = nm.replace("void ", "").replace("(anonymous namespace)::", "")[:64]
    dur[key] += e - s
    n[key] += 1
    if nxt is not None:
        gap_after[key] += max(0, nxt[1] - e)
span = rows[-1][2] - rows[0][1]
busy = sum(dur.values())
print(f"{len(rows)} dispatches over {span / 1e3:.1f} us, device busy {busy / span * 100:.1f} %")
for key in sorted(n, key=lambda k: -dur[k]):
    print(f"{n[key]:8d} x {dur[key] / n[key] / 1e3:8.2f} us  then idle {gap_after[key] / n[key] / 1e3:6.2f} us   {key}")

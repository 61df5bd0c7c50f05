import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
for key in ("update_E", "update_H"):
    rows = [r for r in csv.DictReader(open(f)) if key in r["Kernel_Name"]]
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
    print(key, "launches", len(d), "min", min(d), "max", max(d), "first8", d[:8], "grid", rows[0]["Grid_Size"], rows[0]["Kernel_Name"][:70])

#!/usr/bin/env python3
"""Per-kernel timeline of the LAST inner iteration in a rocprofv3 --kernel-trace CSV (default: newest under gpurun_out/prof)."""
import csv
import glob
import sys

f = sys.argv[1] if len(sys.argv) > 1 else sorted(glob.glob("gpurun_out/prof/runc/*_kernel_trace.csv"))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "cp_update" in r["Kernel_Name"] or "adam_update" in r["Kernel_Name"]]
a, b = idx[-2] + 1, idx[-1] + 1
tot = 0.0
t_first = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    n = r["Kernel_Name"].replace("void pl::", "").replace("_ZN2pl", "")[:52]
    print("%9.1f us  grid %6d  %s" % (d, int(r["Grid_Size_X"]) // 256, n))
print("sum of kernel durations %.1f us; span %.1f us" % (tot, (int(rows[b - 1]["End_Timestamp"]) - t_first) / 1e3))

#!/usr/bin/env python3
"""Timeline of the last planning iteration out of a rocprofv3 --kernel-trace CSV: every kernel with start/end (us, relative to the
iteration's first kernel), so that the critical path across hardware queues can be read off.
usage: timeline.py DIR_OR_CSV [first_kernel_substring]   (default anchor: pack_cp, the first launch of an iteration)"""
import csv
import glob
import os
import sys

src = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else "pack_cp"
if os.path.isdir(src):
    src = sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True))[0]
rows = list(csv.DictReader(open(src)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
if len(starts) < 2:
    sys.exit(f"anchor {anchor!r} seen {len(starts)} times")
a, b = starts[-2], starts[-1]      # the last COMPLETE iteration
t0 = int(rows[a]["Start_Timestamp"])
print(f"# {src}: iteration = rows {a}..{b - 1}, {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us from anchor to anchor")
for r in rows[a:b]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("paule_hip::", "")[:60]
    print(f"{s:10.1f} {e:10.1f} {e - s:9.1f}  q{r.get('Queue_Id', '?'):>3} grid {r.get('Grid_Size', r.get('Grid_Size_X', '?')):>7}  {name}")

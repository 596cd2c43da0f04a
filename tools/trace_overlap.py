#!/usr/bin/env python3
"""Reads a rocprofv3 kernel-trace CSV and reports how much of the sweep kernels' time overlaps: trace_overlap.py <dir or csv>"""
import csv
import glob
import os
import sys

path = sys.argv[1]
files = [path] if os.path.isfile(path) else glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
sw = [r for r in rows if "sweep" in r[2]]
print(f"{len(rows)} kernels, {len(sw)} sweep launches, queues {sorted(set(r[3] for r in rows))}, streams {sorted(set(r[4] for r in rows))}")
ev = []
for s, e, *_ in sw:
    ev.append((s, 1))
    ev.append((e, -1))
ev.sort()
depth, last, hist = 0, None, {}
for t, d in ev:
    if last is not None and depth > 0:
        hist[depth] = hist.get(depth, 0) + (t - last)
    depth += d
    last = t
tot = sum(hist.values())
for k in sorted(hist):
    print(f"  {k} sweeps running: {hist[k] / 1e3:9.1f} us ({100 * hist[k] / max(tot, 1):.1f} %)")
tail = [r for r in rows if r[0] >= sw[-min(len(sw), 30)][0]]
t0 = tail[0][0]
for s, e, n, q, st in tail:
    print(f"  {(s - t0) / 1e3:9.1f} .. {(e - t0) / 1e3:9.1f} us ({(e - s) / 1e3:7.1f})  q{q}  {n[:70]}")

"""Kernel intervals of a rocprofv3 --kernel-trace run (CSV): name, start and end in ms relative to the first kernel of the
last `n` dispatches.   python tools/trace_intervals.py <dir-or-csv> [n]"""
import csv
import glob
import os
import sys

path = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    a, b = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    print(f"{a:8.3f} .. {b:8.3f} ms  ({b - a:7.3f})  queue {r.get('Queue_Id', '?'):>3}  {r['Kernel_Name'][:60]}  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?'))}")

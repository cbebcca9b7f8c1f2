#!/usr/bin/env python3
"""Prints the device timeline of the last batches of a rocprofv3 --kernel-trace run (start offset, duration, gap to the previous
kernel, queue) - for reading launch gaps and overlaps between the chain and the tail stream."""
import csv
import glob
import sys

src = sys.argv[1]
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 40
trace = glob.glob(f"{src}/**/*kernel_trace.csv", recursive=True)
rows = sorted(csv.DictReader(open(trace[0])), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n_last:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][-40:]
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:7.1f}  q{r.get('Queue_Id', '?')}  {name}")
    prev_end = max(prev_end, e)

#!/usr/bin/env python3
"""Prints the device timeline of a rocprofv3 --kernel-trace run (start, end, duration, queue, kernel): how the streaming
kernels, frame kernels and tails of consecutive batches overlap.  By default the window around the last four full-batch
launches of k_frame_lds (its largest grid in the trace); `timeline.py DIR N` prints the last N rows instead."""
import csv
import glob
import sys

src = sys.argv[1]
trace = glob.glob(f"{src}/**/*kernel_trace.csv", recursive=True)
rows = sorted(csv.DictReader(open(trace[0])), key=lambda r: int(r["Start_Timestamp"]))
if len(sys.argv) > 2:
    rows = rows[-int(sys.argv[2]):]
else:
    fr = [i for i, r in enumerate(rows) if "k_frame_lds" in r["Kernel_Name"]]
    gmax = max(int(rows[i]["Grid_Size_X"]) for i in fr)
    fr = [i for i in fr if int(rows[i]["Grid_Size_X"]) == gmax]
    st = [int(rows[i]["Start_Timestamp"]) for i in fr]
    print(f"# k_frame_lds launches with grid {gmax}: {len(fr)}; start-to-start spacing (us):", [round((b - a) / 1e3) for a, b in zip(st, st[1:])])
    rows = rows[max(fr[-5] - 6, 0): fr[-1] + 8] if len(fr) >= 5 else rows[max(fr[0] - 6, 0): fr[-1] + 8]
t0 = int(rows[0]["Start_Timestamp"])
print("#  start us ->    end us  (duration)  queue  kernel")
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][-40:]
    print(f"{(s - t0) / 1e3:9.1f} -> {(e - t0) / 1e3:9.1f}  (+{(e - s) / 1e3:7.1f})  q{r.get('Queue_Id', '?')}  {name}")

#!/usr/bin/env python3
"""Idle time between the frame workgroups that followed each other on one CU.  Input: the file VOFOD_LDS_PROF=2
VOFOD_LDS_PROF_RAW=<file> leaves when the process ends - one line per frame: launch, start us, end us, xcc, se, sh, cu (the
100 MHz wall clock, stamps 0 and 13 of k_frame_lds_far).  usage: tools/cu_gaps.py raw.txt [frames_per_launch]"""
import collections
import json
import sys

import numpy as np

F = int(sys.argv[2]) if len(sys.argv) > 2 else 256
a = np.loadtxt(sys.argv[1])
per_launch = collections.Counter(a[:, 0].astype(int))
keep = sorted(l for l, c in per_launch.items() if c == F)
# the longest stretch of consecutive full launches whose starts follow each other closely: the timed loop
starts = {l: a[a[:, 0] == l][:, 1].min() for l in keep}
runs, cur = [], [keep[0]]
for l0, l1 in zip(keep, keep[1:]):
    if l1 == l0 + 1 and starts[l1] - starts[l0] < 1500:
        cur.append(l1)
    else:
        runs.append(cur)
        cur = [l1]
runs.append(cur)
run = max(runs, key=len)[2:-2]
sel = a[np.isin(a[:, 0].astype(int), run)]
per_cu = collections.defaultdict(list)
for l, s, e, xcc, se, sh, cu in sel:
    per_cu[(int(xcc), int(se), int(sh), int(cu))].append((s, e))
gaps, busy, span, dur = [], 0.0, 0.0, []
for v in per_cu.values():
    v.sort()
    gaps += [s1 - e0 for (s0, e0), (s1, e1) in zip(v, v[1:])]
    busy += sum(e - s for s, e in v)
    span += v[-1][1] - v[0][0]
    dur += [e - s for s, e in v]
g = np.array(gaps)
t_first, t_last = sel[:, 1].min(), sel[:, 2].max()
out = {"launches": len(run), "cus_seen": len(per_cu), "frames": int(len(sel)), "us_per_launch": round((t_last - t_first) / len(run), 1),
       "busy_share_of_cu_time": round(busy / span, 4), "frame_us": {"mean": round(float(np.mean(dur)), 1), "p90": round(float(np.percentile(dur, 90)), 1)},
       "gap_us": {"mean": round(float(g.mean()), 2), "median": round(float(np.median(g)), 2), "p90": round(float(np.percentile(g, 90)), 2), "max": round(float(g.max()), 1),
                  "share_over_20us": round(float((g > 20).mean()), 3)},
       "frames_per_cu": {"min": min(len(v) for v in per_cu.values()), "max": max(len(v) for v in per_cu.values())}}
print(json.dumps(out, indent=1))

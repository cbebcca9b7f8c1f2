#!/usr/bin/env python3
"""Reduces rocprofv3 CSV output (gpurun_out/prof_<tag>/{stats,fetch,write}) to small committed summaries:
profiles/<tag>_kernel_stats.csv (the --stats table) and profiles/<tag>_traffic.json (HBM bytes per launch per
kernel from the FETCH_SIZE / WRITE_SIZE passes, corrected as MI355X_MICROARCH.md §HBM prescribes)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(k_\w+)(<\d+>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:48]


def main():
    src, tag = sys.argv[1], sys.argv[2]
    os.makedirs("profiles", exist_ok=True)
    stats = glob.glob(f"{src}/stats/**/*kernel_stats.csv", recursive=True)
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        with open(f"profiles/{tag}_kernel_stats.csv", "w") as f:
            f.write("kernel,calls,total_ns,avg_ns,pct,min_ns,max_ns\n")
            for r in rows:
                f.write(f"{short(r['Name'])},{r['Calls']},{r['TotalDurationNs']},{float(r['AverageNs']):.0f},{r['Percentage']},{r['MinNs']},{r['MaxNs']}\n")
    # The profiled command warms the map with single-scan launches before the batched steps; the bench line is about
    # the batched launches, so a second table restricts every kernel to its largest grid (= the full-batch launches).
    trace = glob.glob(f"{src}/stats/**/*kernel_trace.csv", recursive=True)
    if trace:
        per = defaultdict(list)
        for r in csv.DictReader(open(trace[0])):
            g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
            per[short(r["Kernel_Name"])].append((g, int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        with open(f"profiles/{tag}_kernel_stats_batch.csv", "w") as f:
            f.write("kernel,grid_threads,calls,avg_ns,min_ns,max_ns\n")
            rows = []
            for k, v in per.items():
                gmax = max(g for g, _ in v)
                d = [t for g, t in v if g == gmax]
                rows.append((sum(d), k, gmax, len(d), sum(d) / len(d), min(d), max(d)))
            for tot, k, gmax, n, avg, mn, mx in sorted(rows, reverse=True):
                f.write(f"{k},{gmax},{n},{avg:.0f},{mn},{mx}\n")
    traffic = defaultdict(lambda: {"launches": 0, "FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0})
    for which, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        for path in glob.glob(f"{src}/{which}/**/*counter_collection.csv", recursive=True):
            rows = [r for r in csv.DictReader(open(path)) if r.get("Counter_Name") == ctr]
            gmax = defaultdict(int)
            for r in rows:
                gmax[short(r["Kernel_Name"])] = max(gmax[short(r["Kernel_Name"])], int(r["Grid_Size"]))
            for r in rows:
                k = short(r["Kernel_Name"])
                if int(r["Grid_Size"]) != gmax[k]:
                    continue  # keep the batched launches only
                traffic[k][ctr] += float(r["Counter_Value"])
                if ctr == "FETCH_SIZE":
                    traffic[k]["launches"] += 1
    out = {}
    for k, t in traffic.items():
        n = max(t["launches"], 1)
        # counters are in KiB; gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide coalesced streaming read (x2 correction,
        # calibrated for 16-B/lane streams only: treat it as an upper estimate for scattered kernels); WRITE_SIZE is exact.
        out[k] = {
            "launches": n,
            "fetch_kib_raw_per_launch": t["FETCH_SIZE"] / n,
            "write_kib_per_launch": t["WRITE_SIZE"] / n,
            "hbm_bytes_per_launch_raw": (t["FETCH_SIZE"] + t["WRITE_SIZE"]) * 1024 / n,
            "hbm_bytes_per_launch_corrected": (2 * t["FETCH_SIZE"] + t["WRITE_SIZE"]) * 1024 / n,
        }
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import kernel_source_sha

    out["kernel_source_sha"] = kernel_source_sha()  # bench.py reports `traffic` only for the sources it was measured on
    json.dump(out, open(f"profiles/{tag}_traffic.json", "w"), indent=1, sort_keys=True)
    print("wrote", f"profiles/{tag}_kernel_stats.csv", f"profiles/{tag}_traffic.json", len(out), "kernels")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Reduces the rocprofv3 --pmc passes of tools/pmc_sq.sh to profiles/<tag>_sq_counters.json: per batch kernel (its launches with
the largest grid: the 256-frame batches) the raw counters per launch and what DESIGN.md quotes from them - per-wave cycle
shares (waiting / issue-stalled / issuing), instructions per wave by class, LDS bank-conflict share.
SQ_WAVE_CYCLES and the SQ_WAIT / SQ_ACTIVE counters tick in quad-cycles (4 clocks) summed over waves."""
import collections
import csv
import glob
import json
import re
import sys


def short(name):
    m = re.search(r"(k_\w+)", name)
    return m.group(1) if m else name[:40]


def main():
    src, tag = sys.argv[1], sys.argv[2]
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    grids = collections.defaultdict(int)
    rows = []
    for sub in ("a", "b", "c"):
        for f in glob.glob(f"{src}/{sub}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                g = int(r["Grid_Size"])
                grids[k] = max(grids[k], g)
                rows.append((k, g, r["Counter_Name"], float(r["Counter_Value"]), int(r.get("Workgroup_Size", 0) or 0)))
    wg = {}
    for k, g, c, v, w in rows:
        if g == grids[k]:
            per[k][c].append(v)
            wg[k] = w
    want = ("k_key1", "k_frame_lds", "k_tail_far", "k_explore", "k_tail_prep", "k_raycast", "k_ray_sweep", "k_mapbits", "k_dilate")
    out = {"_note": "counters per launch (mean over the launches with the kernel's largest grid); *_cycles in quad-cycles summed over waves"}
    for k in want:
        if k not in per:
            continue
        c = {n: sum(v) / len(v) for n, v in per[k].items()}
        e = {"grid_threads": grids[k], "workgroup": wg.get(k), "launches": len(next(iter(per[k].values()))), "counters": {n: round(v, 1) for n, v in sorted(c.items())}}
        waves = c.get("SQ_WAVES")
        wc = c.get("SQ_WAVE_CYCLES")
        if wc:
            e["shares_of_wave_cycles"] = {
                "waiting (s_waitcnt / barrier)": round(c.get("SQ_WAIT_ANY", 0) / wc, 3),
                "issue-stalled": round(c.get("SQ_WAIT_INST_ANY", 0) / wc, 3),
                "issuing": round(c.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3),
                "issuing VALU": round(c.get("SQ_ACTIVE_INST_VALU", 0) / wc, 3),
            }
        if waves:
            e["per_wave"] = {n.replace("SQ_INSTS_", "insts_").lower(): round(c[n] / waves, 1) for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR") if n in c}
            if wc:
                e["per_wave"]["quad_cycles"] = round(wc / waves, 1)
        if c.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_bank_conflict_share"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"], 3)
        out[k] = e
    json.dump(out, open(f"profiles/{tag}_sq_counters.json", "w"), indent=1, sort_keys=True)
    print("wrote", f"profiles/{tag}_sq_counters.json", [k for k in out if not k.startswith("_")])


if __name__ == "__main__":
    main()

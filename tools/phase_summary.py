#!/usr/bin/env python3
"""Reduce the per-launch phase tables (VOFOD_LDS_PROF_JSON) of 256-frame launches to one table: per phase the mean over launches
of the frames' mean / median / p90 / max, alone and in the pipeline.  usage: phase_summary.py alone.jsonl pipe.jsonl"""
import json
import sys


def reduce(path, frames=256, skip=2):
    rows = [json.loads(l) for l in open(path) if l.strip()]
    rows = [r for r in rows if r["frames"] == frames][skip:]
    if not rows:
        return None
    out = {"launches": len(rows), "span_us": round(sum(r["span_us"] for r in rows) / len(rows), 1)}
    for k in ("start_us", "end_us"):
        out[k] = {q: round(sum(r[k][q] for r in rows) / len(rows), 1) for q in rows[0][k]}
    out["phases"] = {}
    for ph in rows[0]["phases"]:
        out["phases"][ph] = {q: round(sum(r["phases"][ph][q] for r in rows) / len(rows), 2) for q in ("mean", "median", "p90", "max")}
    tl = [r for r in rows if "tail" in r]
    if tl:  # k_tail_far's own stamps for the same frames (one wave per frame)
        out["k_tail_far"] = {ph: {q: round(sum(r["tail"][ph][q] for r in tl) / len(tl), 2) for q in ("mean", "median", "p90", "max")} for ph in tl[0]["tail"]}
    return out


res = {"_note": "k_frame_lds_far, 256 x OS1-128 @ 0.25 m: us per phase from the 100 MHz stamps of every frame's workgroup (tools/phase_table.sh); "
                "alone = one batch at a time; pipeline = four batches in flight (k_key1 of the next batch and the tail of the previous one beside it)",
       "alone": reduce(sys.argv[1]), "pipeline": reduce(sys.argv[2])}
print(json.dumps(res, indent=1))

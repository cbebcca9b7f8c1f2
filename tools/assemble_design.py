#!/usr/bin/env python3
"""Writes the measured numbers of a round into DESIGN.md's placeholders ({{NAME}}) from the files under profiles/ - so that
the text quotes what the committed profiles say.  usage: tools/assemble_design.py DESIGN.in.md > DESIGN.md"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
P = ROOT / "profiles"
TAG = "r05"

b = json.loads((P / f"{TAG}_bench_default.json").read_text())
drv = [json.loads(l) for l in (P / f"{TAG}_driver_cmd_repeat.jsonl").read_text().splitlines() if l.strip()]
tr = json.loads((P / f"{TAG}_traffic.json").read_text())
sq = json.loads((P / f"{TAG}_sq_counters.json").read_text())
ph = json.loads((P / f"{TAG}_frame_phases.json").read_text())
cg = json.loads((P / f"{TAG}_cu_gaps.json").read_text())
cfg = [json.loads(l) for l in (P / f"{TAG}_configs_2_3_5.jsonl").read_text().splitlines() if l.strip()]

k = lambda x: f"{x / 1e3:.0f} k"
r = b["roofline"]
fr = tr.get("k_frame_lds", {})
fr_mb = fr.get("hbm_bytes_per_launch_corrected", 0) / 1e6
fetch_mb = fr.get("fetch_kib_raw_per_launch", 0) * 1024 * 2 / 1e6
write_mb = fr.get("write_kib_per_launch", 0) * 1024 / 1e6
must = r["alg_bytes_per_launch"] / 1e6
sqf = sq["k_frame_lds"]["shares_of_wave_cycles"]
drv_s = " / ".join(f"{d['value'] / 1e3:.0f} k" for d in drv)
s32 = b["config3_share"]
lt = b["loaded_tail"]
ss = b["single_stream"]
tail_ph = ph["pipeline"].get("k_tail_far", {})

rows = [
    ("`bench.py` default (200 steps) / the driver's command (`--steps 20 --warmup 5`, three runs)", "835 k / 787-791 k (driver record: 769 k)",
     f"**{b['value']:,.0f}** frames/s, {b['ms_per_step']:.4f} ms per 256 frames (`profiles/{TAG}_bench_default.json`) / **{drv_s}** (`profiles/{TAG}_driver_cmd_repeat.jsonl`); 930-975 k over the round's boxes in 40-200 step runs".replace(",", " ")),
    ("kernels of a batch, alone (HIP events in the bench line / rocprofv3 `--inflight 1`)", "`k_key1` 116 + `k_frame_lds_far` 175-180 + `k_tail_far` 130 µs",
     f"`k_frame_lds_far` {b['kernels']['k_frame_lds_far']['avg_us']:.1f} µs (rocprofv3: `profiles/{TAG}_kernel_stats_batch.csv`) + `k_tail_far` {b['kernels']['k_tail_far']['avg_us']:.1f} µs; no `k_key1`"),
    ("contract path `12·N + 40·V` = 746.6 MB over the kernel sum / the pipelined step", "31.4 % / 30.4 %",
     f"**{100 * r['path']['frac']:.1f} %** / **{100 * r['path']['pipelined']['frac']:.1f} %** of 8 TB/s ({r['path']['pipelined']['GBps'] / 1e3:.2f} TB/s over the step)"),
    ("bytes the close-first path must move, 543.2 MB", "22.9 % / 22.1 %", f"{100 * r['path_moved']['frac']:.1f} % / {100 * r['path_moved']['pipelined']['frac']:.1f} %; the dominant kernel on ITS must-move bytes ({must:.0f} MB - it now reads the input itself): **{100 * r['frac']:.1f} %**"),
    ("PMC traffic of the frame kernel", "529 MB (+ 474 MB of `k_key1`) = 1 003 MB for the two",
     f"**{fr_mb:.0f} MB** ({fetch_mb:.0f} MB fetched with the guide's × 2 correction, {write_mb:.0f} MB written; `profiles/{TAG}_traffic.json`) = {fr_mb / must:.2f} × its {must:.0f} MB"),
    ("SQ counters of the frame kernel (`profiles/%s_sq_counters.json`)" % TAG, "14.3 k VALU per wave, 57 % waiting / 20 % issue-stalled / 23 % issuing, 37 % LDS bank conflicts",
     f"{sq['k_frame_lds']['per_wave']['insts_valu'] / 1e3:.1f} k VALU + {sq['k_frame_lds']['per_wave']['insts_salu'] / 1e3:.1f} k SALU per wave (the input pass is inside now), {100 * sqf['waiting (s_waitcnt / barrier)']:.0f} % waiting / {100 * sqf['issue-stalled']:.0f} % issue-stalled / {100 * sqf['issuing']:.0f} % issuing, {100 * sq['k_frame_lds']['lds_bank_conflict_share']:.0f} % LDS bank conflicts"),
    ("CU time (`tools/cu_gaps.py`, frames laid end to end per CU)", "-",
     f"CUs busy with frames {100 * cg['busy_share_of_cu_time']:.0f} % of the time; gap between two frames of a CU: median {cg['gap_us']['median']:.0f} µs, mean {cg['gap_us']['mean']:.0f} µs, {100 * cg['gap_us']['share_over_20us']:.0f} % over 20 µs (`profiles/{TAG}_cu_gaps.json`)"),
    ("`k_tail_far` per frame in the pipeline (its own stamps)", "-",
     (f"boxes + gates {tail_ph['boxes_gates']['mean']:.0f} µs, flood fills {tail_ph['flood_fills']['mean']:.0f} µs, total {tail_ph['total']['mean']:.0f} µs (34 / 95 / 129 µs before the round's last change)" if tail_ph else "-")),
    ("32-frame batches (configs[3]'s per-GPU share), eight in flight", "179-232 k", f"{k(s32['frames_per_s'])} frames/s ({s32['ms_per_step']:.3f} ms per batch; host {s32.get('host_us_per_submit', 0):.0f} µs per submit; 310-393 k over the round's boxes)"),
    ("loaded tail (12 targets, 751 detections per step)", "503 k", f"{k(lt['frames_per_s'])} frames/s (557-648 k over the round's boxes)"),
    ("host-resident columns / 48-byte structs", "34.5 k / 8.1 k", f"{b['host_input']['frames_per_s'] / 1e3:.1f} k ({b['host_input']['h2d_GBps']:.1f} GB/s) / {b['host_input_aos']['frames_per_s'] / 1e3:.1f} k ({b['host_input_aos']['h2d_GBps']:.1f} GB/s)"),
    ("single sensor stream", "0.184-0.21 ms per scan, 14 launches", f"{ss['ms_per_scan']:.3f} ms per scan in the bench line (0.181 in `tools/single_stream_profile.py`, `profiles/{TAG}_single_stream_kernels.txt`), 14 launches"),
    ("a whole sensor period (configs[1] / [2] / [4], `profiles/%s_configs_2_3_5.jsonl`)" % TAG, "0.62 / 0.67 / 2.8 ms", " / ".join(f"{c['ms_per_scan_full_cycle']:.2f}" for c in cfg) + " ms; `k_raycast` " + " / ".join(f"{c['kernels']['k_raycast']['avg_us']:.0f}" for c in cfg) + " µs; `k_ray_sweep` " + " / ".join(f"{100 * c['roofline']['k_ray_sweep']['frac_of_8TBps']:.0f} %" for c in cfg) + ", `k_mapbits` " + " / ".join(f"{100 * c['roofline']['k_mapbits']['frac_of_8TBps']:.0f} %" for c in cfg) + " of 8 TB/s"),
    ("CPU restatement, 1 core", "75.0 frames/s", f"{b['cpu_baseline']['value']:.1f} frames/s"),
]
table = "| | round 4 | round 5 |\n|---|---|---|\n" + "\n".join(f"| {a} | {o} | {n} |" for a, o, n in rows)

sub = {
    "MEASURED_TABLE": table,
    "TRAFFIC_RESULT": f"{fr_mb:.0f} MB per launch = {fr_mb / must:.2f} × the {must:.0f} MB it must move (round 4: 529 MB for the frame kernel + 474 MB for `k_key1` in front of it)",
    "HEADLINE": f"{100 * r['path']['pipelined']['frac']:.1f} % on the contract's bytes over the pipelined step of the 200-step run, {k(b['value'])} frames/s; with the driver's 20-step command {drv_s}",
    "DRIVER": f"{drv_s} with the driver's command",
    "TRAFFIC_SQ": f"frame kernel {fr_mb:.0f} MB ({fr_mb / must:.2f} × its must-move bytes; the ≤ 300 MB asked for is below the 402 MB of input the kernel now reads itself), waiting share {100 * sqf['waiting (s_waitcnt / barrier)']:.0f} % (asked: < 45 %)",
    "SHARE32": f"{k(s32['frames_per_s'])} frames/s with eight chains in flight on this box (310-393 k over the round's boxes); the host spends {s32.get('host_us_per_submit', 0):.0f} µs in a submit, the chain of a batch is `k_init_hdr` 6 + `k_frame_lds_far` 218 + `k_tail_far` ~130 µs = ~355 µs of latency, so eight in flight can deliver at most 8 × 32 / 355 µs = 720 k; the tail's latency (flood fills) is the part to cut",
    "SHARE32_STATUS": f"**not met**: {k(s32['frames_per_s'])} frames/s (round 4: 246 k in the driver's line); `host_us_per_submit` / `_per_collect` are in the line now ({s32.get('host_us_per_submit', 0):.0f} / {s32.get('host_us_per_collect', 0):.0f} µs, the latter mostly waiting); one launch less per batch (`k_key1` gone); no graph replay, no multi-batch submit",
    "LOADED": f"{k(lt['frames_per_s'])} frames/s on this box (round 4: 494 k; 557-648 k over the round's boxes): workgroup-scope fences, eight frames per tail workgroup, fill loads issued together, member centres in LDS; the four-waves-per-frame tail of the verdict was not built",
}
text = Path(sys.argv[1]).read_text()
for name, val in sub.items():
    text = text.replace("{{" + name + "}}", val)
assert "{{" not in text, [l for l in text.splitlines() if "{{" in l][:3]
sys.stdout.write(text)

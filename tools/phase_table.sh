#!/bin/bash
# Per-phase table of k_frame_lds_far (VERDICT r4 #1a): the 100 MHz stamps of every frame's workgroup, alone (one batch at a time,
# the stream synchronised behind every launch) and in the pipeline (four batches in flight, stamps read when a batch is collected).
# usage (on the GPU box): tools/phase_table.sh <tag>   ->  gpurun_out/<tag>/phases_{alone,pipe}.jsonl
TAG=${1:-ph}
O=gpurun_out/$TAG
mkdir -p $O
rm -f $O/phases_alone.jsonl $O/phases_pipe.jsonl
COMMON="--cpu-baseline-scans 0 --no-profile-pass --host-input-steps 0 --loaded-tail-steps 0"
VOFOD_LDS_PROF=1 VOFOD_LDS_PROF_JSON=$O/phases_alone.jsonl timeout -k 10 300 python bench.py --steps 6 --warmup 2 --inflight 1 $COMMON > $O/prof1.json 2> $O/prof1.err || exit 1
VOFOD_LDS_PROF=2 VOFOD_LDS_PROF_JSON=$O/phases_pipe.jsonl timeout -k 10 300 python bench.py --steps 40 --warmup 8 $COMMON > $O/prof2.json 2> $O/prof2.err || exit 1
python3 tools/phase_summary.py $O/phases_alone.jsonl $O/phases_pipe.jsonl > $O/frame_phases.json && cat $O/frame_phases.json

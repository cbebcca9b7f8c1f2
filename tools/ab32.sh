#!/bin/bash
# A/B of library builds on small batches (configs[3]'s per-GPU share is 32 frames): usage tools/ab32.sh "A:4 B:8" [rounds] [frames]
# (variant = library in ab_libs/ : batches in flight)
VARS=${1:-"A:4 B:4"}
ROUNDS=${2:-3}
FR=${3:-32}
for i in $(seq $ROUNDS); do
  for v in $VARS; do
    lib=${v%%:*}; infl=${v##*:}
    cp ab_libs/$lib.so vofod_amd/csrc/libvofod_hip.so
    timeout -k 10 200 python bench.py --frames $FR --inflight $infl --steps 400 --warmup 30 --cpu-baseline-scans 0 --host-input-steps 0 --no-profile-pass 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value']), round(d['ms_per_step'],4))"
  done
done

#!/bin/bash
# A/B of builds of the library on ONE box (box-to-box variance is larger than most kernel changes):
# ab_libs/<V>.so are copied over the in-tree library in turn, bench runs interleaved.  usage: tools/ab.sh "A B C" [rounds]
VARS=${1:-"A B"}
ROUNDS=${2:-3}
for i in $(seq $ROUNDS); do
  for v in $VARS; do
    cp ab_libs/$v.so vofod_amd/csrc/libvofod_hip.so
    timeout -k 10 200 python bench.py --steps 40 --warmup 8 --cpu-baseline-scans 0 --no-profile-pass --host-input-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value']), round(d['ms_per_step'],3))"
  done
done

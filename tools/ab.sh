#!/bin/bash
# A/B of builds of the library on ONE box (box-to-box variance is larger than most kernel changes):
# ab_libs/<V>.so are copied over the in-tree library in turn, bench runs interleaved.  usage: tools/ab.sh "A B C" [rounds]
# Prints the pipelined rate and the kernels' own durations (HIP events, one batch at a time: the steadier number).
VARS=${1:-"A B"}
ROUNDS=${2:-3}
for i in $(seq $ROUNDS); do
  for v in $VARS; do
    cp ab_libs/$v.so vofod_amd/csrc/libvofod_hip.so
    timeout -k 10 200 python bench.py --steps 40 --warmup 8 --cpu-baseline-scans 0 --host-input-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('$v', round(d['value']), round(d['ms_per_step'],3), {n.split('<')[0]: round(x['avg_us'],1) for n,x in k.items() if n.startswith(('k_key1','k_frame_lds'))})"
  done
done

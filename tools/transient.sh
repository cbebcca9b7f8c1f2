#!/bin/bash
# Diagnostics of the start-up transient of the timed region: the driver's exact command, with the
# completion time of every step.
TAG=${1:-tr}
O=gpurun_out/$TAG; mkdir -p $O
for i in 1 2; do
  VOFOD_BENCH_STEPLOG=1 timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-baseline-scans 0 --no-profile-pass > $O/drv_$i.json 2> $O/drv_$i.err || exit 1
  VOFOD_BENCH_GC=on VOFOD_BENCH_STEPLOG=1 timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-baseline-scans 0 --no-profile-pass > $O/gcon_$i.json 2> $O/gcon_$i.err || exit 1
done
grep -H steps $O/*.err
python3 -c "
import json,glob
for f in sorted(glob.glob('$O/*.json')):
    d=json.load(open(f)); print(f, round(d['value']), round(d['ms_per_step'],3))"

#!/bin/bash
# A/B of environment settings on ONE box with the in-tree library: usage tools/abenv.sh "VOFOD_X=1 VOFOD_X=2" [rounds] [bench args]
VARS=${1}
ROUNDS=${2:-3}
shift 2
for i in $(seq $ROUNDS); do
  for v in $VARS; do
    env $v timeout -k 10 200 python bench.py --steps 100 --warmup 10 --cpu-baseline-scans 0 --host-input-steps 0 --no-profile-pass "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value']), round(d['ms_per_step'],4))"
  done
done

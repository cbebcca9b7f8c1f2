#!/bin/bash
# device timeline of pipelined 32-frame batches (configs[3]'s per-GPU share) under rocprofv3
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-t32}
OUT=$R/gpurun_out/trace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/p -- python3 $R/bench.py --frames 32 --max-batch 256 --steps 12 --warmup 3 --cpu-baseline-scans 0 --no-profile-pass --host-input-steps 0 > $OUT/p.log 2>&1 || { tail -5 $OUT/p.log; exit 1; }
cd $R && python3 tools/timeline.py $OUT/p 140 > $OUT/timeline.txt; tail -90 $OUT/timeline.txt

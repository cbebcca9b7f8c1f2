#!/bin/bash
# device timeline of the pipelined default workload (256-frame batches, 3 in flight) under rocprofv3
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-pipe}
OUT=$R/gpurun_out/trace_$TAG
mkdir -p $OUT
# the library keeps a dozen streams busy: bench.py asks for 16 hardware queues, but under rocprofv3 the runtime is initialised
# before python starts - the variable has to come from this shell
export GPU_MAX_HW_QUEUES=16
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/p -- python3 $R/bench.py --steps 12 --warmup 3 --cpu-baseline-scans 0 --no-profile-pass --host-input-steps 0 --loaded-tail-steps 0 > $OUT/p.log 2>&1 || { tail -5 $OUT/p.log; exit 1; }
cd $R && python3 tools/timeline.py $OUT/p > $OUT/timeline.txt; head -50 $OUT/timeline.txt

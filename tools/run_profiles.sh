#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel stats + separate PMC passes of the default bench workload.
# Raw output goes to gpurun_out/ (scratch); tools/summarize_profiles.py reduces it into profiles/ (committed).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r03}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
# the library keeps a dozen streams busy: bench.py asks for 16 hardware queues, but under rocprofv3 the runtime is initialised
# before python starts - the variable has to come from this shell
export GPU_MAX_HW_QUEUES=16
cd /tmp && export TMPDIR=/tmp
# --inflight 1: one batch at a time, so that a kernel's duration and its counters are its own (in the pipelined default the
# streaming kernels of batch k+1 run beside the frame kernel of batch k: durations stretch, and the device-wide PMC
# counters of a dispatch include its neighbours' traffic)
CMD="python3 $R/bench.py --steps 5 --warmup 1 --inflight 1 --cpu-baseline-scans 0 --no-profile-pass --host-input-steps 0 --loaded-tail-steps 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD > $OUT/fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD > $OUT/write.log 2>&1 || exit 1
# and the pipelined default, for the record of how the kernels overlap (tools/timeline.py reads it)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pipelined -- python3 $R/bench.py --steps 12 --warmup 2 --cpu-baseline-scans 0 --no-profile-pass --host-input-steps 0 --loaded-tail-steps 0 > $OUT/pipelined.log 2>&1 || exit 1
cd $R && python3 tools/summarize_profiles.py $OUT $TAG && python3 tools/timeline.py $OUT/pipelined > profiles/${TAG}_timeline_pipelined.txt

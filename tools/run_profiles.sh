#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel stats + separate PMC passes of the default bench workload.
# Raw output goes to gpurun_out/ (scratch); tools/summarize_profiles.py reduces it into profiles/ (committed).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r02}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 5 --warmup 1 --cpu-baseline-scans 0 --no-profile-pass"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD > $OUT/fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD > $OUT/write.log 2>&1 || exit 1
cd $R && python3 tools/summarize_profiles.py $OUT $TAG

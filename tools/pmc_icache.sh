#!/bin/bash
# instruction-fetch counters of the batch kernels, one batch at a time (k_frame_lds is ~300 KB of code behind a 64 KB cache)
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-ic}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
# the library keeps a dozen streams busy: bench.py asks for 16 hardware queues, but under rocprofv3 the runtime is initialised
# before python starts - the variable has to come from this shell
export GPU_MAX_HW_QUEUES=16
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 4 --warmup 1 --inflight 1 --cpu-baseline-scans 0 --no-profile-pass --host-input-steps 0"
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/a -- $CMD > $OUT/a.log 2>&1 || { tail -5 $OUT/a.log; exit 1; }
cd $R && python3 - <<PY
import csv,glob,collections
f=glob.glob("$OUT/a/**/*counter_collection.csv",recursive=True)
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k=r["Kernel_Name"].split("(")[0][-28:]
    if int(r["Grid_Size"]) < 65536: continue
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    print(k,{c:round(sum(x)/len(x)/1e6,3) for c,x in v.items()}, "(M per launch)")
PY

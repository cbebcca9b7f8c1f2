#!/bin/bash
# SQ counters of the batch kernels, one batch at a time (wave-cycle breakdown: parked / issue-stalled / issuing)
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-sq}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
# the library keeps a dozen streams busy: bench.py asks for 16 hardware queues, but under rocprofv3 the runtime is initialised
# before python starts - the variable has to come from this shell
export GPU_MAX_HW_QUEUES=16
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 4 --warmup 1 --inflight 1 --cpu-baseline-scans 0 --no-profile-pass --host-input-steps 0"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/a -- $CMD > $OUT/a.log 2>&1 || { tail -5 $OUT/a.log; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/b -- $CMD > $OUT/b.log 2>&1 || { tail -5 $OUT/b.log; exit 1; }
cd $R && python3 - <<PY
import csv,glob,collections
for sub in ("a","b"):
    f=glob.glob("$OUT/"+sub+"/**/*counter_collection.csv",recursive=True)
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k=r["Kernel_Name"].split("(")[0][-28:]
        if int(r["Grid_Size"]) < 65536: continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items():
        print(sub,k,{c:round(sum(x)/len(x)/1e6,2) for c,x in v.items()}, "(M per launch)")
PY

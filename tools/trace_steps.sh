#!/bin/bash
# kernel trace of a bench run with enough steps to see the steady state; prints the spacing of the batch chains
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/trace_${1:-x}
mkdir -p $OUT
# the library keeps a dozen streams busy: bench.py asks for 16 hardware queues, but under rocprofv3 the runtime is initialised
# before python starts - the variable has to come from this shell
export GPU_MAX_HW_QUEUES=16
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/bench.py --steps 40 --warmup 5 --cpu-baseline-scans 0 --no-profile-pass ${BENCH_ARGS:-} > $OUT/log.txt 2>&1 || exit 1
cd $R && python3 - <<PY
import csv,glob
rows=sorted(csv.DictReader(open(glob.glob('$OUT/**/*kernel_trace.csv',recursive=True)[0])),key=lambda r:int(r['Start_Timestamp']))
fr=[i for i,r in enumerate(rows) if 'k_frame_lds' in r['Kernel_Name'] and int(r['Grid_Size_X'])==256*1024]
st=[int(rows[i]['Start_Timestamp']) for i in fr]
d=[(b-a)/1e3 for a,b in zip(st,st[1:])]
print('chain spacing us:', [round(x) for x in d[8:40]])
i0=fr[20]-9; i1=fr[23]+6
t0=int(rows[i0]['Start_Timestamp']); prev=t0
for r in rows[i0:i1]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    print(f"{(s-t0)/1e3:9.1f} +{(e-s)/1e3:7.1f} gap {(s-prev)/1e3:7.1f} q{r['Queue_Id']} {r['Kernel_Name'].split('(')[0][-30:]}")
    prev=max(prev,e)
PY

#!/bin/bash
# device timeline of pipelined 32-frame batches (configs[3]'s per-GPU share) under rocprofv3: the steady-state middle of the run
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-t32}
OUT=$R/gpurun_out/trace_$TAG
mkdir -p $OUT
# the library keeps a dozen streams busy: bench.py asks for 16 hardware queues, but under rocprofv3 the runtime is initialised
# before python starts - the variable has to come from this shell
export GPU_MAX_HW_QUEUES=16
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/p -- python3 $R/bench.py --frames 32 --steps 60 --warmup 10 --cpu-baseline-scans 0 --no-profile-pass --host-input-steps 0 > $OUT/p.log 2>&1 || { tail -5 $OUT/p.log; exit 1; }
cd $R && python3 - $OUT <<'PY' > $OUT/timeline.txt
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/p/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_frame_lds" in r["Kernel_Name"]]
i0 = idx[len(idx) // 2]
t0 = int(rows[i0]["Start_Timestamp"])
names = ("k_frame_lds", "k_key1", "k_tail_prep", "k_explore", "k_tail_finish", "k_init_hdr", "copyBuffer")
for r in rows[i0 - 3 : i0 + 80]:
    n = r["Kernel_Name"]
    short = [x for x in names if x in n]
    s = (int(r["Start_Timestamp"]) - t0) / 1e3
    e = (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{s:9.1f} -> {e:9.1f} (+{e - s:7.1f}) q{r['Queue_Id']} {short[0] if short else n[:30]}")
PY
cat $OUT/timeline.txt

#!/bin/bash
# SQ counters of the batch kernels, one batch at a time (wave-cycle breakdown: parked / issue-stalled / issuing), reduced to
# profiles/<tag>_sq_counters.json (tracked: DESIGN.md's "instruction- and latency-bound" statements are recomputable from it).
# Usage on the GPU box: bash tools/pmc_sq.sh r04
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-sq}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
# the library keeps a dozen streams busy: bench.py asks for 16 hardware queues, but under rocprofv3 the runtime is initialised
# before python starts - the variable has to come from this shell
export GPU_MAX_HW_QUEUES=16
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 4 --warmup 1 --inflight 1 --cpu-baseline-scans 0 --no-profile-pass --host-input-steps 0 --loaded-tail-steps 0"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/a -- $CMD > $OUT/a.log 2>&1 || { tail -5 $OUT/a.log; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/b -- $CMD > $OUT/b.log 2>&1 || { tail -5 $OUT/b.log; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/c -- $CMD > $OUT/c.log 2>&1 || { tail -5 $OUT/c.log; exit 1; }
cd $R && python3 tools/summarize_sq.py $OUT $TAG

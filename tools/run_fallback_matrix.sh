#!/bin/bash
# GPU parity suite under every switch that turns a fast path off (README "Environment switches"); one line per run.
# Usage on the GPU box: bash tools/run_fallback_matrix.sh [quick|core]   (quick: the switches of the batched fast path only;
# core: every switch, but only the parity / KAT / close-first files - a minute per switch instead of five)
set -o pipefail
ALL=("" VOFOD_CLOSE_FIRST=0 VOFOD_DEVICE_TAIL=0 VOFOD_LITE=0 VOFOD_SLABS=0 VOFOD_SLAB_EMIT=0 VOFOD_BRICK_LDS=0 VOFOD_DILATE=0 VOFOD_CCL=voxel "VOFOD_DEVICE_TAIL=0 VOFOD_LITE=0" "VOFOD_CLOSE_FIRST=0 VOFOD_DEVICE_TAIL=0" VOFOD_EXPLORE=host)
QUICK=(VOFOD_CLOSE_FIRST=0 VOFOD_DEVICE_TAIL=0 "VOFOD_DEVICE_TAIL=0 VOFOD_LITE=0" VOFOD_BRICK_LDS=0 VOFOD_CCL=voxel)
if [ "$1" = quick ]; then SW=("${QUICK[@]}"); else SW=("${ALL[@]}"); fi
TESTS=tests
if [ "$1" = core ]; then TESTS="tests/test_gpu_parity.py tests/test_gpu_kat.py tests/test_gpu_close_first.py"; fi
for sw in "${SW[@]}"; do
  printf "%-36s " "${sw:-default}"
  env $sw timeout -k 10 900 python -m pytest $TESTS -x -q -m gpu 2>&1 | grep -E "^FAILED|passed|failed|error" | tail -3 | tr "\n" " "; echo
done

#!/bin/bash
# round-5 experiment call: raycast variants, input-prefetch A/B, batches in flight
O=gpurun_out/c3; mkdir -p $O
cp vofod_amd/csrc/libvofod_hip.so $O/orig.so
for v in R0 R1 R2 R2N R2S; do
  cp ab_libs/$v.so vofod_amd/csrc/libvofod_hip.so
  echo "== $v os1-128 0.25"; timeout -k 10 120 python tools/ray_probe.py os1-128 0.25 2>&1 | tail -3
done > $O/ray.log 2>&1
for v in R0 R2; do
  cp ab_libs/$v.so vofod_amd/csrc/libvofod_hip.so
  echo "== $v os2 0.1"; timeout -k 10 200 python tools/ray_probe.py os2-128x2048 0.1 2>&1 | tail -3
done >> $O/ray.log 2>&1
cat $O/ray.log
tools/ab.sh "R2 P2" 3 > $O/ab_prefetch.log 2>&1; cat $O/ab_prefetch.log
cp ab_libs/P2.so vofod_amd/csrc/libvofod_hip.so
for n in 4 6 8; do
  echo "inflight $n"; timeout -k 10 200 python bench.py --steps 100 --warmup 10 --cpu-baseline-scans 0 --host-input-steps 0 --no-profile-pass --loaded-tail-steps 0 --inflight $n 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],4))"
done > $O/inflight.log 2>&1; cat $O/inflight.log
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "raycast or kat or stream_route or close_first or bench_workload or sequence_parity" > $O/tests_P2.log 2>&1; tail -3 $O/tests_P2.log
cp $O/orig.so vofod_amd/csrc/libvofod_hip.so

#!/bin/bash
O=gpurun_out/c5; mkdir -p $O
cp vofod_amd/csrc/libvofod_hip.so $O/orig.so
for v in X1 X2; do
  cp ab_libs/$v.so vofod_amd/csrc/libvofod_hip.so
  echo "== $v os1-128 0.25"; timeout -k 10 120 python tools/ray_probe.py os1-128 0.25 2>&1 | tail -3
done > $O/ray.log 2>&1
cp ab_libs/X1.so vofod_amd/csrc/libvofod_hip.so
echo "== X1 os2 0.1" >> $O/ray.log; timeout -k 10 250 python tools/ray_probe.py os2-128x2048 0.1 2>&1 | tail -3 >> $O/ray.log
cat $O/ray.log
timeout -k 10 500 python -m pytest tests -x -q -m gpu -k "raycast or kat or stream_route or sequence_parity or apriori_map" > $O/tests.log 2>&1; tail -3 $O/tests.log
cp $O/orig.so vofod_amd/csrc/libvofod_hip.so

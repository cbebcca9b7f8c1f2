#!/bin/bash
O=gpurun_out/c6; mkdir -p $O
cp vofod_amd/csrc/libvofod_hip.so $O/orig.so
for v in A0 A3 A4; do
  cp ab_libs/$v.so vofod_amd/csrc/libvofod_hip.so
  echo "== $v os1-128 0.25"; timeout -k 10 120 python tools/ray_probe.py os1-128 0.25 2>&1 | tail -3
done > $O/ray.log 2>&1
cat $O/ray.log
cp $O/orig.so vofod_amd/csrc/libvofod_hip.so

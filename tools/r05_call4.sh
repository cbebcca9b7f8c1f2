#!/bin/bash
# round-5 experiment call 4: raycast variants (labelled), tail workgroup size A/B
O=gpurun_out/c4; mkdir -p $O
cp vofod_amd/csrc/libvofod_hip.so $O/orig.so
for v in T4 R1 R4 R2N R2S; do
  cp ab_libs/$v.so vofod_amd/csrc/libvofod_hip.so
  echo "== $v os1-128 0.25"; timeout -k 10 120 python tools/ray_probe.py os1-128 0.25 2>&1 | tail -3
done > $O/ray.log 2>&1
cat $O/ray.log
tools/ab.sh "T4 T8 T16" 3 > $O/ab_tail.log 2>&1; cat $O/ab_tail.log
for v in T4 T16; do
cp ab_libs/$v.so vofod_amd/csrc/libvofod_hip.so
echo "== $v loaded tail + 32-frame"; timeout -k 10 200 python bench.py --steps 60 --warmup 8 --cpu-baseline-scans 0 --host-input-steps 0 --no-profile-pass 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), 'loaded', round(d['loaded_tail']['frames_per_s']), 'share32', round(d['config3_share']['frames_per_s']), 'single', round(d['single_stream']['ms_per_scan'],4))"
done > $O/legs.log 2>&1; cat $O/legs.log
cp $O/orig.so vofod_amd/csrc/libvofod_hip.so

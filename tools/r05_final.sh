#!/bin/bash
# Round-5 final measurements on the GPU box (final code): everything judged lands in gpurun_out/r05_summary/ (copy to profiles/)
TAG=r05
S=gpurun_out/${TAG}_summary; mkdir -p $S
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_gpu_tests.log 2>&1; tail -3 gpurun_out/${TAG}_gpu_tests.log | tee $S/${TAG}_gpu_tests_tail.txt
tools/run_profiles.sh ${TAG} > gpurun_out/${TAG}_profiles.log 2>&1; tail -2 gpurun_out/${TAG}_profiles.log
cp profiles/${TAG}_kernel_stats*.csv profiles/${TAG}_traffic.json profiles/${TAG}_timeline_pipelined.txt $S/
timeout -k 10 400 python bench.py > $S/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err
python -c "
import json;d=json.load(open('$S/${TAG}_bench_default.json'));r=d['roofline'];print(round(d['value']),round(d['ms_per_step'],4),round(r['frac'],4),r['traffic'],round(r['path']['frac'],4),round(r['path']['pipelined']['frac'],4),d['single_stream']['ms_per_scan'],d['loaded_tail']['frames_per_s'],d['config3_share']['frames_per_s'])"
rm -f $S/${TAG}_driver_cmd_repeat.jsonl
for i in 1 2 3; do timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null >> $S/${TAG}_driver_cmd_repeat.jsonl; done
python -c "
import json
for l in open('$S/${TAG}_driver_cmd_repeat.jsonl'): d=json.loads(l); print('driver cmd', round(d['value']), round(d['ms_per_step'],4), d['roofline']['traffic'])"
tools/phase_table.sh ${TAG}ph > gpurun_out/${TAG}_phase.log 2>&1; cp gpurun_out/${TAG}ph/frame_phases.json $S/${TAG}_frame_phases.json; tail -12 gpurun_out/${TAG}_phase.log

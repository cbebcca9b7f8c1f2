#!/bin/bash
# one gpurun call: GPU parity suite, then short benches (frames per batch from $SWEEP)
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -${TAILN:-2} || exit 1
for F in ${SWEEP:-32}; do
timeout -k 10 300 python bench.py --steps 30 --warmup 3 --cpu-baseline-scans 0 --frames $F > gpurun_out/bench_F$F.json 2> gpurun_out/bench_F$F.err && python -c "
import json,sys;d=json.load(open('gpurun_out/bench_F$F.json'));print('F=$F',round(d['value']),round(d['ms_per_step'],3),round(d['single_stream']['ms_per_scan'],3)); print({k:round(v['avg_us'],1) for k,v in d['kernels'].items()})"
done

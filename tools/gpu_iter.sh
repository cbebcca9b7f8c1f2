#!/bin/bash
# one gpurun call of the optimisation loop: GPU parity suite (fail fast), phase stamps of the frame kernel, bench line
TAG=${1:-x}
timeout -k 10 800 python -m pytest tests -q -m gpu -x --durations=3 > gpurun_out/${TAG}_test.log 2>&1; tail -6 gpurun_out/${TAG}_test.log
VOFOD_LDS_PROF=1 timeout -k 10 300 python bench.py --steps 2 --warmup 1 --cpu-baseline-scans 0 --no-profile-pass > gpurun_out/${TAG}_prof.json 2> gpurun_out/${TAG}_prof.err; grep k_frame gpurun_out/${TAG}_prof.err | tail -4
timeout -k 10 300 python bench.py --cpu-baseline-scans 0 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python -c "
import json;d=json.load(open('gpurun_out/${TAG}_bench.json'));print(round(d['value']),round(d['ms_per_step'],3)); print({k:round(v['avg_us'],1) for k,v in d['kernels'].items()})"

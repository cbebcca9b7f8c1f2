#!/bin/bash
# The measurement suite of a round (run on the GPU box): bench line, rocprofv3 summaries, the other configurations, fuzz.
TAG=${1:-r02}
set -o pipefail
timeout -k 10 400 python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err && cp gpurun_out/${TAG}_bench_default.json profiles/${TAG}_bench_default.json
python -c "
import json;d=json.load(open('gpurun_out/${TAG}_bench_default.json'));print(round(d['value']),round(d['ms_per_step'],3),d['roofline']['frac'],d['roofline']['path']['frac'],d['cpu_baseline']['value'],d.get('config3_share'))"
tools/run_profiles.sh ${TAG} > gpurun_out/${TAG}_profiles.log 2>&1; tail -2 gpurun_out/${TAG}_profiles.log
timeout -k 10 600 python tools/bench_configs.py > gpurun_out/${TAG}_configs.jsonl 2> gpurun_out/${TAG}_configs.err && cp gpurun_out/${TAG}_configs.jsonl profiles/${TAG}_configs_2_3_5.jsonl
tail -c 600 gpurun_out/${TAG}_configs.jsonl

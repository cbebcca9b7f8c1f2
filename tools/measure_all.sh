#!/bin/bash
# The measurement suite of a round (run on the GPU box): rocprofv3 summaries first (tools/run_profiles.sh writes
# profiles/<tag>_traffic.json for the kernel sources at hand), then the bench line (which picks the PMC traffic up), then
# the other configurations.  Everything judged is also copied under gpurun_out/<tag>_summary/ - the only directory that
# comes back from the box; copy it into profiles/ afterwards.
TAG=${1:-r03}
set -o pipefail
mkdir -p gpurun_out/${TAG}_summary
tools/run_profiles.sh ${TAG} > gpurun_out/${TAG}_profiles.log 2>&1; tail -2 gpurun_out/${TAG}_profiles.log
cp profiles/${TAG}_kernel_stats*.csv profiles/${TAG}_traffic.json profiles/${TAG}_timeline_pipelined.txt gpurun_out/${TAG}_summary/
timeout -k 10 400 python bench.py > gpurun_out/${TAG}_summary/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err
python -c "
import json;d=json.load(open('gpurun_out/${TAG}_summary/${TAG}_bench_default.json'));r=d['roofline'];print(round(d['value']),round(d['ms_per_step'],3),round(r['frac'],4),r['traffic'],round(r['path']['frac'],4),round(r['path']['pipelined']['frac'],4),d['cpu_baseline']['value'],d.get('config3_share'))"
timeout -k 10 600 python tools/bench_configs.py > gpurun_out/${TAG}_summary/${TAG}_configs_2_3_5.jsonl 2> gpurun_out/${TAG}_configs.err
tail -c 600 gpurun_out/${TAG}_summary/${TAG}_configs_2_3_5.jsonl

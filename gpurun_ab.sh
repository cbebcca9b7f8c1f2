python -m pytest tests -x -q -m gpu 2>&1 | tail -2
VOFOD_TRACE=1 python bench.py --steps 6 --warmup 2 --cpu-baseline-scans 0 --no-profile-pass > gpurun_out/bench_t.json 2> gpurun_out/bench_t.err; grep "n=32" gpurun_out/bench_t.err | tail -2; python -c "
import json;d=json.load(open('gpurun_out/bench_t.json'));print(round(d['value']),round(d['ms_per_step'],3),d['single_stream']['ms_per_scan'])"

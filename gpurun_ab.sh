for w in 48 96; do python bench.py --steps 5 --warmup 1 --map-warm-scans $w --cpu-baseline-scans 0 --no-profile-pass > gpurun_out/bench_w$w.json 2> gpurun_out/bench_w$w.err; python -c "
import json;d=json.load(open('gpurun_out/bench_w$w.json'));print('warm $w',round(d['value']),d['config']['detections_per_step'])"; done

python -m pytest tests -x -q -m gpu 2>&1 | tail -2
python bench.py --steps 30 --warmup 3 --cpu-baseline-scans 0 > gpurun_out/bench3.json 2> gpurun_out/bench3.err; python -c "
import json;d=json.load(open('gpurun_out/bench3.json'));print(round(d['value']),round(d['ms_per_step'],3),round(d['single_stream']['ms_per_scan'],3)); print({k:round(v['avg_us'],1) for k,v in d['kernels'].items()})"

python -m pytest tests -x -q -m gpu 2>&1 | tail -2
timeout -k 10 600 python tools/bench_configs.py 2 5 > gpurun_out/configs_b.jsonl 2> gpurun_out/configs_b.err; python -c "
import json
for l in open('gpurun_out/configs_b.jsonl'):
    d=json.loads(l); print(d['config'], round(d['scans_per_s_full_cycle'],1), 'scans/s', round(d['ms_per_scan_full_cycle'],2), 'ms'); print('  ', {k:v['avg_us'] for k,v in d['kernels'].items() if v['avg_us']>30}); print('  ', {k:round(v['frac_of_8TBps'],3) for k,v in d.get('roofline',{}).items()})
"
python bench.py --steps 10 --warmup 2 --cpu-baseline-scans 0 > gpurun_out/bench3.json 2> gpurun_out/bench3.err; python -c "
import json;d=json.load(open('gpurun_out/bench3.json'));print(round(d['value']),round(d['ms_per_step'],2),round(d['single_stream']['ms_per_scan'],3)); print({k:round(v['avg_us'],1) for k,v in d['kernels'].items()})"

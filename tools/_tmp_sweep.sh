for rep in 1 2 3; do for inf in 2 3; do
timeout -k 10 300 python bench.py --steps 30 --warmup 3 --cpu-baseline-scans 0 --no-profile-pass --inflight $inf > gpurun_out/s.json 2> gpurun_out/s.err && python -c "
import json;d=json.load(open('gpurun_out/s.json'));print('inflight=$inf',round(d['value']),round(d['ms_per_step'],3))"
done; done
VOFOD_TRACE=1 timeout -k 10 300 python bench.py --steps 6 --warmup 2 --cpu-baseline-scans 0 --no-profile-pass > gpurun_out/s.json 2> gpurun_out/s.err; grep "n=256" gpurun_out/s.err | tail -3

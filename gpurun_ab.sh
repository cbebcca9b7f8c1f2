for u in 2 3; do cp gpurun_lib_ufm$u.so vofod_amd/csrc/libvofod_hip.so; for m in 1 3; do VOFOD_BRICK_MODE=$m python bench.py --steps 20 --warmup 3 --cpu-baseline-scans 0 > gpurun_out/bench_b.json 2> gpurun_out/bench_b.err; python -c "
import json;d=json.load(open('gpurun_out/bench_b.json'));print('ufm $u mode $m',round(d['value']),round(d['ms_per_step'],3)); print({k:round(v['avg_us'],1) for k,v in d['kernels'].items() if 'brick' in k})"; done; done
cp gpurun_lib_ufm3.so vofod_amd/csrc/libvofod_hip.so; python -m pytest tests -x -q -m gpu 2>&1 | tail -2

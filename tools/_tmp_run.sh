VOFOD_LDS_PROF=1 timeout -k 10 300 python bench.py --steps 2 --warmup 1 --cpu-baseline-scans 0 --no-profile-pass > gpurun_out/p.json 2> gpurun_out/p.err; grep "k_slab_emit" gpurun_out/p.err | tail -2

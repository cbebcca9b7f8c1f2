import os, sys, time, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, torch
torch.cuda.init()
import vofod_amd
from vofod_amd import capi, synth
from vofod_amd.detector import ScanData, VoFOD, default_params
lib = vofod_amd.library()
h, w, vfov, _ = synth.SENSORS["os1-128"]
sp, dp = default_params(lib); sp.voxel_size = 0.25; sp.sensor_hrays, sp.sensor_vrays = w, h; sp.sensor_vfov = np.float32(np.deg2rad(vfov)); sp.max_batch_frames = 4
det = VoFOD(lib, sp, dp)
scene = synth.bench_scene(); synth.warm_map(det, scene, "os1-128", 24)
seq = synth.scan_sequence(scene, "os1-128", 12, seed0=5000)
dev = torch.device("cuda", 0); keep = []; sd = []
for s in seq:
    t = torch.from_numpy(np.stack([s.x, s.y, s.z])).to(dev); keep.append(t)
    sd.append(ScanData(x=t[0].data_ptr(), y=t[1].data_ptr(), z=t[2].data_ptr(), width=w, height=h, stride_bytes=4, memspace=capi.MEM_DEVICE))
torch.cuda.synchronize()
det.process_scan(sd[0], seq[0].tf)
t1 = time.perf_counter()
for s, d in zip(seq[1:], sd[1:]): det.process_scan(d, s.tf)
print("ms per scan", 1e3 * (time.perf_counter() - t1) / (len(seq) - 1))
lib.profile_enable(det.h, 1)
for s, d in zip(seq[1:6], sd[1:6]): det.process_scan(d, s.tf)
names, ms, calls = (C.c_char * (64 * 96))(), (C.c_double * 96)(), (C.c_uint64 * 96)()
n = lib.profile_read(det.h, names, ms, calls, 96)
tot = 0
for i in range(n):
    nm = names[64 * i:64 * i + 64].split(b"\0", 1)[0].decode(); us = 1e3 * ms[i] / max(calls[i], 1); tot += us * calls[i] / 5
    print(f"{nm:28s} {us:7.1f} us x {calls[i] / 5:.1f}")
print("sum per scan (us)", tot)

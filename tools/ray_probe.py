#!/usr/bin/env python3
"""diagnostics: k_raycast time for one OS1-128 scan at 0.25 m (VOFOD_RAY_SKIPNEAR leaves the near field out)"""
import ctypes as C, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import vofod_amd
from vofod_amd import capi, synth
from vofod_amd.detector import VoFOD, default_params
lib = vofod_amd.library()
sensor, vs = sys.argv[1] if len(sys.argv) > 1 else "os1-128", float(sys.argv[2]) if len(sys.argv) > 2 else 0.25
h, w, vfov, _ = synth.SENSORS[sensor]
sp, dp = default_params(lib)
sp.voxel_size = vs; sp.sensor_hrays, sp.sensor_vrays = w, h; sp.sensor_vfov = np.float32(np.deg2rad(vfov))
det = VoFOD(lib, sp, dp)
scene = synth.make_scene(0, n_targets=3)
synth.seed_ground(det)
scans = synth.scan_sequence(scene, sensor, 6, seed0=1000)
lib.profile_enable(det.h, 1)
for k, s in enumerate(scans):
    det.process_scan(s.scan, s.tf)
    if k + 1 < len(scans):
        det.raycast_begin(s.scan, s.tf)
        det.process_scan(scans[k + 1].scan, scans[k + 1].tf)
        det.raycast_finish(allow=(capi.ERR_RAYCAST_NO_DETECTION, capi.ERR_RAYCAST_EMPTY))
names = (C.c_char * (64 * 96))(); ms = (C.c_double * 96)(); calls = (C.c_uint64 * 96)()
n = lib.profile_read(det.h, names, ms, calls, 96)
for i in range(n):
    nm = names[64 * i: 64 * i + 64].split(b"\0", 1)[0].decode()
    if "ray" in nm:
        print(nm, round(1e3 * ms[i] / max(calls[i], 1), 1), "us", int(calls[i]))

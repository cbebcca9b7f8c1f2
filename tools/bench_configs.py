#!/usr/bin/env python3
"""Measures the BASELINE.json configurations that bench.py's contract line does not cover (run on the GPU box):
  config 2: OS1-128, 0.25 m, single sensor stream: process_scan + raycast + sepclusters roles, sequential
  config 3: config 2 + apriori (+inf) background map, classification and flood fill active
  config 5: OS2-128 2048 columns, 0.1 m voxels (M = 301.7 M), weighted grid + raycast + update sweep
Prints one JSON line per configuration with per-kernel times from the library's HIP-event profiler."""
import ctypes as C
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

import vofod_amd  # noqa: E402
from vofod_amd import capi, synth  # noqa: E402
from vofod_amd.detector import ScanData, VoFOD, default_params  # noqa: E402


def device_scan(s, sensor, dev):
    h, w, _, _ = synth.SENSORS[sensor]
    t = torch.from_numpy(np.stack([s.x, s.y, s.z, s.intensity, s.range.view(np.float32)])).to(dev)
    return t, ScanData(x=t[0].data_ptr(), y=t[1].data_ptr(), z=t[2].data_ptr(), intensity=t[3].data_ptr(), range=t[4].data_ptr(), width=w, height=h,
                       stride_bytes=4, memspace=capi.MEM_DEVICE)


def kernels_of(lib, det):
    names = (C.c_char * (64 * 96))()
    ms = (C.c_double * 96)()
    calls = (C.c_uint64 * 96)()
    n = lib.profile_read(det.h, names, ms, calls, 96)
    return {names[64 * i : 64 * i + 64].split(b"\0", 1)[0].decode(): {"avg_us": 1e3 * ms[i] / max(calls[i], 1), "launches": int(calls[i])} for i in range(n)}


def run(name, sensor, vs, apriori, n_warm, n_timed):
    lib = vofod_amd.library()
    dev = torch.device("cuda", 0)
    h, w, vfov_deg, _ = synth.SENSORS[sensor]
    sp, dp = default_params(lib)
    sp.voxel_size = vs
    sp.sensor_hrays, sp.sensor_vrays = w, h
    sp.sensor_vfov = np.float32(np.deg2rad(vfov_deg))
    det = VoFOD(lib, sp, dp)
    scene = synth.make_scene(0, n_targets=3)
    if apriori:
        det.load_apriori(synth.apriori_points(scene, vs, n_voxels=1_000_000, solid_ground_to=-1.2))
    else:
        # range-finder stand-in without a host round trip of the whole map: apriori-style ground disc, then reset latches by hand
        gx, gy = np.meshgrid(np.arange(-20, 30, vs), np.arange(-20, 30, vs), indexing="ij")
        pts = np.stack([gx.ravel(), gy.ravel(), np.full(gx.size, 0.01)], axis=1).astype(np.float32)
        pts = pts[np.hypot(pts[:, 0], pts[:, 1]) < 30]
        det.load_apriori(pts)
    scans = synth.scan_sequence(scene, sensor, n_warm + n_timed + 1, seed0=1000)
    keep, dscans = [], []
    for s in scans:
        t, d = device_scan(s, sensor, dev)
        keep.append(t)
        dscans.append(d)
    torch.cuda.synchronize()

    def cycle(k):
        # one sensor period: detection, raycast of this scan finishing after the next detection iteration, sepclusters every 2nd scan
        dets = det.process_scan(dscans[k], scans[k].tf, flags=capi.SCAN_AUTO_RAYCAST)
        if k % 2 == 1:
            st, sure = det.sepclusters_begin(allow=(capi.ERR_EMPTY,))
            if st == capi.OK and sure:
                det.sepclusters_finish()
        return len(dets)

    for k in range(n_warm):
        cycle(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nd = 0
    for k in range(n_warm, n_warm + n_timed):
        nd += cycle(k)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    lib.profile_enable(det.h, 1)
    for k in range(n_warm + n_timed - 4, n_warm + n_timed):
        cycle(k)
    kern = kernels_of(lib, det)
    lib.profile_enable(det.h, 0)
    _, dbg = det.process_scan(dscans[-1], scans[-1].tf, flags=capi.SCAN_NO_MAP_UPDATE, debug=True)
    M = det.n_voxels
    out = {
        "config": name, "sensor": sensor, "voxel_size": vs, "points_per_scan": h * w, "map_voxels": M, "voxels_per_scan": len(dbg["weighted"]),
        "scans_per_s_full_cycle": n_timed / dt, "ms_per_scan_full_cycle": 1e3 * dt / n_timed, "detections": nd,
        "kernels": {k: {"avg_us": round(v["avg_us"], 1), "launches": v["launches"]} for k, v in kern.items()},
    }
    # HBM-bound sweeps: algorithmic bytes / time
    # k_ray_sweep reads flags[i] and ray[i] of every voxel (8 bytes) and touches map[i] only where a ray passed: 8*M is its
    # algorithmic traffic (+ 8 bytes per updated voxel and the clears, a few percent) - not the 12*M round 1 quoted
    for k, b in (("k_ray_sweep", 8.0 * M), ("k_mapbits", 4.0 * M + M / 8.0)):
        if k in kern and kern[k]["avg_us"] > 0:
            out.setdefault("roofline", {})[k] = {"alg_bytes": b, "GBps": b / (kern[k]["avg_us"] * 1e-6) / 1e9, "frac_of_8TBps": b / (kern[k]["avg_us"] * 1e-6) / 8e12}
    print(json.dumps(out), flush=True)
    det.close()


if __name__ == "__main__":
    which = sys.argv[1:] or ["2", "3", "5"]
    if "2" in which:
        run("configs[1] single stream", "os1-128", 0.25, False, 24, 12)
    if "3" in which:
        run("configs[2] apriori map", "os1-128", 0.25, True, 12, 12)
    if "5" in which:
        run("configs[4] OS2-128x2048 @0.1 m", "os2-128x2048", 0.1, False, 6, 6)

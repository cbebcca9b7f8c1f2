#!/usr/bin/env python3
"""Stress of the pipelined batch path (runs on the GPU box): the same F-frame batch submitted N times with four tickets in
flight (eight below 128 frames) - every collect must return exactly the detections of the first one (overlapping frame
kernels on two streams and the shared tail stream at 128 frames and more; whole chains with tails and flood-fill buffers of
their own below; the per-ticket workspaces).  usage: stress_pipeline.py [frames] [steps]"""
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")  # (as bench.py: one process owns the GPU)
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

import vofod_amd  # noqa: E402
from vofod_amd import capi, synth  # noqa: E402
from vofod_amd.detector import ScanData, VoFOD, default_params  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
lib = vofod_amd.library()
h, w, vfov, _ = synth.SENSORS["os1-128"]
sp, dp = default_params(lib)
sp.voxel_size = 0.25
sp.sensor_hrays, sp.sensor_vrays = w, h
sp.sensor_vfov = np.float32(np.deg2rad(vfov))
sp.max_batch_frames = F
det = VoFOD(lib, sp, dp)
warm = synth.make_scene(0, n_targets=0)
scene = synth.make_scene(0, n_targets=12)
synth.warm_map(det, warm, "os1-128", 24)
frames = synth.bench_frames(scene, "os1-128", F)
dev = torch.device("cuda", 0)
cols = torch.empty((F, 3, h * w), dtype=torch.float32, device=dev)
for f, s in enumerate(frames):
    cols[f, 0], cols[f, 1], cols[f, 2] = torch.from_numpy(s.x), torch.from_numpy(s.y), torch.from_numpy(s.z)
torch.cuda.synchronize()
scans = [ScanData(x=cols[f, 0].data_ptr(), y=cols[f, 1].data_ptr(), z=cols[f, 2].data_ptr(), width=w, height=h, stride_bytes=4, memspace=capi.MEM_DEVICE) for f in range(F)]
tfs = np.stack([s.tf for s in frames]).astype(np.float32)
want, want_per = det.process_batch(scans, tfs)
assert len(want) >= (50 if F >= 256 else 4), len(want)
depth = 4 if F >= 128 else 8
infl, bad = [], 0
for k in range(steps):
    infl.append(det.batch_submit(scans, tfs))
    if len(infl) == depth:
        got, per = det.batch_collect(infl.pop(0))
        ok = np.array_equal(per, want_per) and len(got) == len(want) and np.array_equal(got["position"], want["position"]) and np.array_equal(got["confidence"], want["confidence"]) and np.array_equal(got["n_points"], want["n_points"])
        bad += 0 if ok else 1
while infl:
    got, per = det.batch_collect(infl.pop(0))
    ok = np.array_equal(per, want_per) and len(got) == len(want) and np.array_equal(got["position"], want["position"]) and np.array_equal(got["confidence"], want["confidence"])
    bad += 0 if ok else 1
print(f"stress: {steps} pipelined batches of {F} frames, {len(want)} detections each, {bad} differing from the synchronous result")
sys.exit(1 if bad else 0)

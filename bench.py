#!/usr/bin/env python3
"""bench.py — LiDAR frames/s of the per-scan hot path on MI355X (BASELINE.json metric).

A step = one batch of F independent synthetic OS1-128 scans (131 072 points each, 0.25 m voxels:
BASELINE.json configs[1] in the batched form of configs[3]) through vofod_batch_submit/collect against a
pre-warmed voxel map, inputs already resident in HBM.  With N GPUs every rank processes its own F frames per
step (weak scaling, no data-path collective; `--scaling strong` splits F frames over the ranks instead) and
the fixed-size detection records are all-gathered with RCCL at the end of each step.  Prints ONE JSON line
on rank 0.

`python bench.py --gpus N` without a launcher starts the N ranks itself (one process per GPU, rendezvous on 127.0.0.1)
before anything touches the GPU; under `python -m torch.distributed.run` the launcher's RANK / WORLD_SIZE are used.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))


HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6290 GB/s is the measured copy ceiling


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="a step lasts ~0.5 ms")
    ap.add_argument("--warmup", type=int, default=10, help="untimed steps (first-use work of the runtime: queues, code objects)")
    ap.add_argument("--frames", type=int, default=256, help="independent scans per GPU per step (one workgroup per frame clusters in LDS: 256 frames fill the 256 CUs)")
    ap.add_argument("--max-batch", type=int, default=0, help="frame slots of the handle's workspaces (default: --frames); spare slots let small batches split their frames into slabs")
    ap.add_argument("--voxel-size", type=float, default=0.25)
    ap.add_argument("--sensor", default="os1-128")
    ap.add_argument("--map-warm-scans", type=int, default=96)
    ap.add_argument("--cpu-baseline-scans", type=int, default=512, help="scans timed through the CPU oracle (0 disables); ~10 s of single-thread CPU work")
    ap.add_argument("--no-profile-pass", action="store_true")
    ap.add_argument("--loaded-tail-steps", type=int, default=40, help="steps of the loaded-tail leg (a scene with 12 floating targets: >= 50 detections per step); 0 disables")
    ap.add_argument("--host-input-steps", type=int, default=12, help="steps of the host-resident input leg (pinned host columns, VOFOD_MEM_HOST: the nodelet's operating point); 0 disables")
    ap.add_argument("--inflight", type=int, default=0, help="batches in flight (1..8; 0 = four for batches of 128 frames and more - the submission of batch k+1 then never waits for the tail of batch k-2, more gain nothing - and eight for smaller batches, whose whole chains run side by side); their kernel chains run on streams of their own and overlap on the device")
    ap.add_argument("--collective", choices=("torch", "cabi"), default="torch", help="N > 1: all-gather through torch.distributed (RCCL / gloo) or through the product's C-ABI (vofod_allgather_detections: RCCL from libvofod_hip.so)")
    ap.add_argument("--backend", default="nccl", help="process-group backend; gloo (CPU tensors) is for rehearsing the N>1 path on a 1-GPU box")
    ap.add_argument("--rehearse-one-gpu", action="store_true", help="every rank uses cuda:0 (only with --backend gloo)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak", help="weak: --frames per GPU per step; strong: --frames in total per step, split over the GPUs (configs[3]: 256 scans over 8 GPUs)")
    return ap.parse_args()


def build_detector(lib, sensor, voxel_size, frames, device):
    from vofod_amd import synth
    from vofod_amd.detector import VoFOD, default_params

    h, w, vfov_deg, _ = synth.SENSORS[sensor]
    sp, dp = default_params(lib)
    sp.voxel_size = voxel_size
    sp.sensor_hrays, sp.sensor_vrays = w, h
    sp.sensor_vfov = np.float32(np.deg2rad(vfov_deg))
    sp.max_batch_frames = frames
    sp.device = device
    return VoFOD(lib, sp, dp)


def launch_ranks(n: int) -> int:
    """`bench.py --gpus N` run as a plain script: start the N ranks as child processes of this one (which never initialises
    the GPU) with the torch.distributed environment of a one-node job; rank 0 prints the JSON line on the shared stdout."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + sys.argv[1:], env=env))
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.05)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in live:  # a rank failed: the others would wait in the rendezvous / a collective for ever
                    q.terminate()
    return rc


def stub_rank(args):
    """VOFOD_BENCH_STUB=1 (tests/test_bench_launch.py, CPU): the launch and rendezvous logic of the N>1 path without the
    GPU workload - every rank joins a gloo group, the ranks' ids are all-gathered, rank 0 prints the line."""
    import torch
    import torch.distributed as dist

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    if not args.rehearse_one_gpu:
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")  # (as main() does before anything initialises the GPU)
    seen = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(seen, torch.tensor([rank, int(os.environ.get("GPU_MAX_HW_QUEUES", "0"))], dtype=torch.int64))
    dist.barrier()
    if rank == 0:
        print(json.dumps({"stub": True, "n_gpus": world, "gpus_arg": args.gpus, "ranks_seen": [int(t[0].item()) for t in seen], "hw_queues_seen": [int(t[1].item()) for t in seen],
                          "local_rank": int(os.environ["LOCAL_RANK"])}))
    dist.destroy_process_group()


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args.gpus))  # (before any GPU initialisation: the children own the devices)
    if os.environ.get("VOFOD_BENCH_STUB") == "1":
        return stub_rank(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not args.rehearse_one_gpu:
        # The library keeps up to a dozen HIP streams busy (key / frame / tail stages, one chain per small batch in flight); the
        # runtime deals a process's streams onto FOUR hardware queues by default, and streams that share a queue take turns (32-frame
        # batches: 131 k -> 227 k frames/s from this variable alone; 256-frame batches: no difference).  Read by the runtime when it
        # initialises, i.e. before the first HIP call of the process.  Every rank that OWNS its GPU asks for them
        # (round 4: a real multi-GPU run used to keep the default of four and would have run its 32-frame share at the ~140 k
        # frames/s of four queues); only the gloo rehearsal of the N > 1 path on ONE card keeps the default: two processes with
        # sixteen queues each take turns on the hardware's queue slots - 54 k instead of 382 k frames/s.
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if args.rehearse_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # RCCL on ROCm
        else:
            dist.init_process_group(args.backend)
    cdev = dev if args.backend == "nccl" else torch.device("cpu")  # where the collective's tensors live

    import vofod_amd
    from vofod_amd import capi, synth
    from vofod_amd.detector import ScanData

    lib = vofod_amd.library()
    F = args.frames if args.scaling == "weak" else max(4, args.frames // world)
    if args.inflight <= 0:
        args.inflight = 4 if F >= 128 else 8
    det = build_detector(lib, args.sensor, args.voxel_size, max(F, args.max_batch), local_rank)
    det.reserve(args.inflight)  # workspaces of the batches in flight: allocated here, not inside the first (warm-up) submits (7 ms each)
    scene = synth.bench_scene()
    synth.warm_map(det, scene, args.sensor, args.map_warm_scans)

    # F independent frames per rank (seeds differ per rank), resident in HBM as packed SoA columns
    h, w, _, _ = synth.SENSORS[args.sensor]
    n_pts = h * w
    def resident(frames_):
        """a set of frames as packed SoA columns in HBM"""
        c = torch.empty((len(frames_), 3, n_pts), dtype=torch.float32, device=dev)
        for f, s in enumerate(frames_):
            c[f, 0] = torch.from_numpy(s.x)
            c[f, 1] = torch.from_numpy(s.y)
            c[f, 2] = torch.from_numpy(s.z)
        sc = [ScanData(x=c[f, 0].data_ptr(), y=c[f, 1].data_ptr(), z=c[f, 2].data_ptr(), width=w, height=h, stride_bytes=4, memspace=capi.MEM_DEVICE) for f in range(len(frames_))]
        return c, sc, np.stack([s.tf for s in frames_]).astype(np.float32)

    host_scans = synth.bench_frames(scene, args.sensor, F, rank)
    cols, scans, tfs = resident(host_scans)
    # a second, different set of F frames (other poses, other noise): the timed loop alternates between the two, so that no step
    # finds its 402 MB of input in the 256 MB Infinity Cache from the step before
    host_scans_b = synth.bench_frames(scene, args.sensor, F, rank + 5000)
    cols_b, scans_b, tfs_b = resident(host_scans_b)
    torch.cuda.synchronize()
    input_sets = [(scans, tfs), (scans_b, tfs_b)]

    from vofod_amd import dist as vdist

    # all-gather payload: D_MAX 128-byte detection records + the count per frame, RCCL over xGMI
    def gather_bufs(n_frames):
        return {
            "local": torch.zeros((n_frames, vdist.FRAME_F64), dtype=torch.float64, device=cdev),
            "all": torch.zeros((world, n_frames, vdist.FRAME_F64), dtype=torch.float64, device=cdev),
            # (two pinned slots: the async copy of one step is not overwritten by the next)
            "hosts": [torch.zeros((n_frames, vdist.FRAME_F64), dtype=torch.float64).pin_memory() for _ in range(2)] if world > 1 else [],
            "count": 0,
        }

    gb = {"cur": gather_bufs(F)}
    cabi_comm = None
    if world > 1 and args.collective == "cabi":
        cabi_comm = vdist.CabiComm(lib, rank, world, local_rank, bootstrap=vdist.torch_bootstrap(cdev))

    def publish(dets, per):
        if cabi_comm is not None:
            cabi_comm.allgather(dets, per)
        elif world > 1:
            b = gb["cur"]
            rec_host = b["hosts"][b["count"] & 1]
            b["count"] += 1
            vdist.pack_detections(dets, per, out=rec_host.numpy())
            b["local"].copy_(rec_host, non_blocking=True)
            vdist.allgather_detections(b["local"], b["all"])

    def run_steps(k, scans=None, tfs=None):
        """k batches through the submit/collect pipeline: batch i+1 is enqueued before batch i is collected, so the host
        tail of one batch overlaps the device chain of the next.  Every batch is submitted and collected inside the call."""
        n_det = 0
        inflight = []
        host_prof = os.environ.get("VOFOD_BENCH_HOSTPROF") == "1"  # diagnostics: where the host thread spends a step
        step_log = [] if os.environ.get("VOFOD_BENCH_STEPLOG") == "1" else None  # diagnostics: completion time of every step
        t_sub = t_col = 0.0
        for i in range(k):
            ta = time.perf_counter()
            sc_i, tf_i = (scans, tfs) if scans is not None else input_sets[i & 1]
            inflight.append(det.batch_submit(sc_i, tf_i))
            tb = time.perf_counter()
            if len(inflight) == args.inflight:
                dets, per = det.batch_collect(inflight.pop(0))
                publish(dets, per)
                n_det += len(dets)
            t_sub += tb - ta
            t_col += time.perf_counter() - tb
            if step_log is not None:
                step_log.append((time.perf_counter(), tb - ta))
        if host_prof and k > 1:
            print(f"[host] per step: submit {1e6 * t_sub / k:.0f} us, collect (incl. waiting) {1e6 * t_col / k:.0f} us", file=sys.stderr)
        while inflight:
            dets, per = det.batch_collect(inflight.pop(0))
            publish(dets, per)
            n_det += len(dets)
        if step_log:
            d = np.diff(np.array([t for t, _ in step_log])) * 1e3
            print(f"[steps] k={k} ms between step completions (ms inside submit): " + " ".join(f"{x:.2f}({1e3 * sb:.2f})" for x, (_, sb) in zip(d, step_log[1:])), file=sys.stderr)
        return n_det

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # The timed loops are a few hundred Python calls into the C library; a cyclic-garbage collection of the interpreter (torch
    # and numpy have loaded hundreds of thousands of objects) in the middle of one costs ~40 ms - seventy steps' worth.  The
    # harness's collector is parked for the measurement; nothing below creates reference cycles.
    import gc

    gc.collect()
    gc.freeze()
    if os.environ.get("VOFOD_BENCH_GC") != "on":  # (tools/transient.sh compares both)
        gc.disable()
    if args.warmup:
        run_steps(args.warmup)
    sync()
    t0 = time.perf_counter()
    n_det = run_steps(args.steps)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    gc.enable()  # the timed region is over: the legs below (and the CPU baseline) run with the collector as usual

    # configs[3] as written: 256 scans per step in total, split over the ranks (strong scaling), all-gather included.  On one
    # GPU this is the headline run itself; on N GPUs every rank takes the first 256 / N of its frames.
    strong = None
    if world > 1 and args.scaling == "weak":
        Fs = max(4, min(F, 256 // world))
        s_scans, s_tfs = scans[:Fs], tfs[:Fs]
        gb["cur"] = gather_bufs(Fs)
        run_steps(max(args.warmup, 3), s_scans, s_tfs)
        sync()
        t1 = time.perf_counter()
        run_steps(args.steps, s_scans, s_tfs)
        sync()
        dts = time.perf_counter() - t1
        t = torch.tensor([dts], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dts = float(t.item())
        strong = {"frames_total_per_step": world * Fs, "frames_per_gpu_per_step": Fs, "frames_per_s": world * Fs * args.steps / dts, "ms_per_step": 1e3 * dts / args.steps,
                  "note": "configs[3]: 256 scans per step split over the GPUs (strong scaling), detections all-gathered every step"}

    # proof that the collective saw every rank (the first real multi-GPU record should show N ranks over RCCL, not N replicas)
    ranks_seen, rccl_version = [0], None
    if world > 1:
        rs = [torch.zeros(1, dtype=torch.int64, device=cdev) for _ in range(world)]
        dist.all_gather(rs, torch.tensor([rank], dtype=torch.int64, device=cdev))
        ranks_seen = [int(t.item()) for t in rs]
        if args.backend == "nccl":
            try:
                rccl_version = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:  # noqa: BLE001
                rccl_version = "unknown"
    out = None
    if rank == 0:
        frames = world * F * args.steps
        # per-frame statistics for the algorithmic-bytes model B = 12*N + 40*V
        _, dbg = det.process_scan(scans[0], tfs[0], flags=capi.SCAN_NO_MAP_UPDATE, debug=True)
        V = len(dbg["weighted"])
        # ... and for the bytes the close-first path really has to move: the voxels of the far clusters (the only ones clustered)
        _, _, dbg_far = det.process_batch(scans[:4], tfs[:4], debug=True, far_only=True)
        V_far = float(np.mean([int((g["labels"] != capi.LABEL_NONE).sum()) for g in dbg_far]))
        out = {
            "metric": "LiDAR frames/sec (131k-pt OS1-128) at 1/2/4/8 MI355X + HBM roofline %",
            "value": frames / dt,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{args.sensor} {h}x{w} scans, {args.voxel_size} m voxels, batched process_scan (configs[1] scan shape in configs[3]'s batched form)",
                "frames_per_gpu_per_step": F,
                "points_per_frame": n_pts,
                "voxels_per_frame": V,
                "map_voxels": det.n_voxels,
                "map_warm_scans": args.map_warm_scans,
                "detections_per_step": n_det / args.steps,
                "pipeline": f"vofod_batch_submit/collect, {args.inflight} batches in flight on streams of their own; classification tail on the device",
                "gpu_max_hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "0")) or "runtime default (4)",
            },
        }
        if world > 1:
            out["ranks_seen"] = ranks_seen
            out["collective"] = {"caller": args.collective, "backend": args.backend, "rccl_version": rccl_version}
        if strong is not None:
            out["config3_strong"] = strong
        elif world == 1 and F == 256 and args.scaling == "weak":
            out["config3_strong"] = {"frames_total_per_step": F, "frames_per_gpu_per_step": F, "frames_per_s": frames / dt, "ms_per_step": 1e3 * dt / args.steps,
                                     "note": "configs[3] on one GPU is the headline run itself (256 scans per step)"}
        # single-stream (stateful, sequential) latency of the same scan shape
        seq = synth.scan_sequence(scene, args.sensor, 17, seed0=5000)
        seq_dev = []
        keep = []
        for s in seq:
            t = torch.from_numpy(np.stack([s.x, s.y, s.z])).to(dev)
            keep.append(t)
            seq_dev.append(ScanData(x=t[0].data_ptr(), y=t[1].data_ptr(), z=t[2].data_ptr(), width=w, height=h, stride_bytes=4, memspace=capi.MEM_DEVICE))
        torch.cuda.synchronize()
        det.process_scan(seq_dev[0], seq[0].tf)
        t1 = time.perf_counter()
        for s, sd in zip(seq[1:], seq_dev[1:]):
            det.process_scan(sd, s.tf)
        single_ms = 1e3 * (time.perf_counter() - t1) / (len(seq) - 1)
        out["single_stream"] = {"ms_per_scan": single_ms, "frames_per_s": 1e3 / single_ms, "scans": len(seq) - 1,
                                "note": "sequential vofod_process_scan with map update (the reference's own mode: one sensor stream), device-resident input, classification tail on the device"}

        if F > 32 and world == 1:
            # configs[3] spreads 256 scans over 8 GPUs: 32 per GPU and step.  The same handle, batches of 32 frames.
            sub, sub_tfs = scans[:32], tfs[:32]

            # eight in flight: every small batch runs its whole chain, tail included, on its ticket's stream.  (The streams must not
            # share hardware queues for that: GPU_MAX_HW_QUEUES above.  With the runtime's default of four, eight in flight gave
            # 142 k frames/s, with sixteen 226-289 k; round 2's three in flight on the shared tail stream: 101-132 k.)
            depth32 = 8

            host32 = {"submit": 0.0, "collect": 0.0, "n": 0}

            def run32(k):
                infl = []
                for _ in range(k):
                    ta = time.perf_counter()
                    infl.append(det.batch_submit(sub, sub_tfs))
                    tb = time.perf_counter()
                    if len(infl) == depth32:
                        det.batch_collect(infl.pop(0))
                    host32["submit"] += tb - ta
                    host32["collect"] += time.perf_counter() - tb  # (includes waiting for the oldest batch in flight)
                    host32["n"] += 1
                while infl:
                    det.batch_collect(infl.pop(0))

            run32(2 * depth32 + 4)  # (every ticket's workspace and buffers are allocated on first use: all of them before the clock starts)
            torch.cuda.synchronize()
            host32.update(submit=0.0, collect=0.0, n=0)
            t1 = time.perf_counter()
            run32(100)
            torch.cuda.synchronize()
            dt32 = time.perf_counter() - t1
            host_us = {"host_us_per_submit": round(1e6 * host32["submit"] / host32["n"], 1), "host_us_per_collect": round(1e6 * host32["collect"] / host32["n"], 1)}
            lib.profile_enable(det.h, 1)
            for _ in range(4):
                det.process_batch(sub, sub_tfs)
            knames, kms, kcalls = (C.c_char * (64 * 64))(), (C.c_double * 64)(), (C.c_uint64 * 64)()
            kn = lib.profile_read(det.h, knames, kms, kcalls, 64)
            lib.profile_enable(det.h, 0)
            k32 = {knames[64 * i : 64 * i + 64].split(b"\0", 1)[0].decode(): round(1e3 * kms[i] / max(kcalls[i], 1), 1) for i in range(kn)}
            out["config3_share"] = {"frames_per_gpu_per_step": 32, "frames_per_s": 32 * 100 / dt32, "ms_per_step": 1e3 * dt32 / 100, "kernel_us": k32,
                                    "batches_in_flight": depth32, **host_us,
                                    "note": "configs[3]'s per-GPU share (256 scans / 8 GPUs): 32-frame batches on this one GPU, eight in flight, each on a stream (and hardware queue) of its own with its own tail; a frame kernel of 32 workgroups leaves 7/8 of the CUs to the other batches in flight"}
        if world == 1 and args.loaded_tail_steps > 0:
            # the classification tail under load: twelve floating targets, nine of which appeared after the map was warmed
            # (the headline scene's three are part of the background by now: ~3 detections per step leave the tail nearly idle)
            busy = synth.make_scene(synth.BENCH_SCENE_SEED, n_targets=12)
            _, scans_t, tfs_t = resident(synth.bench_frames(busy, args.sensor, F, rank))
            run_steps(3, scans_t, tfs_t)
            sync()
            t1 = time.perf_counter()
            nd = run_steps(args.loaded_tail_steps, scans_t, tfs_t)
            sync()
            dtl = time.perf_counter() - t1
            out["loaded_tail"] = {"frames_per_s": F * args.loaded_tail_steps / dtl, "ms_per_step": 1e3 * dtl / args.loaded_tail_steps, "steps": args.loaded_tail_steps,
                                  "detections_per_step": nd / args.loaded_tail_steps,
                                  "note": "the same pipeline on a scene with 12 floating targets (9 new since the warm-up): dozens of candidate clusters, flood fills and detections per batch"}
        if world == 1 and args.host_input_steps > 0:
            out["host_input"] = host_input_leg(args, det, host_scans, tfs, torch, capi, ScanData, h, w)
            out["host_input_aos"] = host_input_aos_leg(args, det, host_scans, tfs, torch, capi, ScanData, h, w)
        if not args.no_profile_pass:
            out["roofline"], out["kernels"] = profile_pass(lib, det, scans, tfs, n_pts, V, F, V_far)
            # the same bytes over the pipelined step time of this rank (kernels of consecutive batches overlap: the step is
            # shorter than the sum of its kernels)
            for key in ("path", "path_moved"):
                pb = out["roofline"][key]["alg_bytes_per_batch"]
                out["roofline"][key]["pipelined"] = {
                    "us_per_batch": 1e3 * out["ms_per_step"],
                    "GBps": pb / (1e-3 * out["ms_per_step"]) / 1e9,
                    "frac": pb / (1e-3 * out["ms_per_step"]) / 1e9 / HBM_PEAK_GBS,
                }
        if args.cpu_baseline_scans > 0:
            out["cpu_baseline"] = cpu_baseline(args, det, host_scans)
    sync()
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def host_input_leg(args, det, host_scans, tfs, torch, capi, ScanData, h, w):
    """The drop-in's real operating point: the nodelet hands over a host-resident cloud (vofod_nodelet.cpp:882,
    INTEGRATION.md passes VOFOD_MEM_HOST).  The batch's packed x|y|z columns sit in ONE pinned host block; the library
    moves them with one 2-D copy per batch on its streaming stage's stream, so the copy of batch k+1 overlaps the frame
    kernel and the tail of batch k.  PCIe-inclusive rate - reported beside `value`, never as `value`."""
    F = len(host_scans)
    n_pts = h * w
    hcols = torch.empty((F, 3, n_pts), dtype=torch.float32).pin_memory()
    for f, s in enumerate(host_scans):
        hcols[f, 0] = torch.from_numpy(s.x)
        hcols[f, 1] = torch.from_numpy(s.y)
        hcols[f, 2] = torch.from_numpy(s.z)
    hs = [ScanData(x=hcols[f, 0].data_ptr(), y=hcols[f, 1].data_ptr(), z=hcols[f, 2].data_ptr(), width=w, height=h, stride_bytes=4, memspace=capi.MEM_HOST) for f in range(F)]

    def run(k):
        infl = []
        for _ in range(k):
            infl.append(det.batch_submit(hs, tfs))
            if len(infl) == args.inflight:
                det.batch_collect(infl.pop(0))
        while infl:
            det.batch_collect(infl.pop(0))

    run(3)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    run(args.host_input_steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t1
    nbytes = 12.0 * n_pts * F
    return {
        "frames_per_s": F * args.host_input_steps / dt,
        "ms_per_step": 1e3 * dt / args.host_input_steps,
        "steps": args.host_input_steps,
        "h2d_bytes_per_step": nbytes,
        "h2d_GBps": nbytes * args.host_input_steps / dt / 1e9,
        "note": "pinned host x|y|z columns (12 B/point), VOFOD_MEM_HOST, one hipMemcpy2DAsync per batch overlapped with the previous batch's chain; bound by the host link "
        "(PCIe Gen5 x16: 64 GB/s raw, ~50-55 GB/s achievable = ~33-36 k frames/s of 1.57 MB); the 48-byte ouster AoS of the nodelet would move 4x the bytes",
    }


def host_input_aos_leg(args, det, host_scans, tfs, torch, capi, ScanData, h, w):
    """What the nodelet really holds: pcl::PointCloud<ouster_ros::Point> - 48-byte structs (x, y, z at bytes 0 / 4 / 8;
    include/vofod/point_types.h), host resident.  The library moves each frame's block with one copy and reads the columns in
    place at the struct's stride: 4 x the bytes of packed columns over the link."""
    F = len(host_scans)
    n_pts = h * w
    POINT = 48
    arena = torch.zeros((F, n_pts, POINT // 4), dtype=torch.float32).pin_memory()
    for f, s in enumerate(host_scans):
        arena[f, :, 0] = torch.from_numpy(s.x)
        arena[f, :, 1] = torch.from_numpy(s.y)
        arena[f, :, 2] = torch.from_numpy(s.z)
    base = arena.data_ptr()
    hs = [ScanData(x=base + f * n_pts * POINT, y=base + f * n_pts * POINT + 4, z=base + f * n_pts * POINT + 8, width=w, height=h, stride_bytes=POINT, memspace=capi.MEM_HOST) for f in range(F)]
    steps = max(2, args.host_input_steps // 3)

    def run(k):
        infl = []
        for _ in range(k):
            infl.append(det.batch_submit(hs, tfs))
            if len(infl) == 2:
                det.batch_collect(infl.pop(0))
        while infl:
            det.batch_collect(infl.pop(0))

    run(2)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t1
    nbytes = float(POINT) * n_pts * F
    return {
        "frames_per_s": F * steps / dt,
        "ms_per_step": 1e3 * dt / steps,
        "steps": steps,
        "h2d_bytes_per_step": nbytes,
        "h2d_GBps": nbytes * steps / dt / 1e9,
        "note": "pinned host array of 48-byte ouster_ros::Point structs (the nodelet's own cloud), VOFOD_MEM_HOST, one copy per frame, columns read in place at stride 48; "
        "PCIe ceiling for this layout: ~50-55 GB/s / 6.3 MB = ~8-9 k frames/s",
    }


def profile_pass(lib, det, scans, tfs, n_pts, V, F, V_far=0.0):
    """Per-kernel device time measured with HIP events on the library's own stream (vofod_profile_*)."""
    from vofod_amd import capi

    lib.profile_enable(det.h, 1)
    reps = 5
    for _ in range(reps):
        det.process_batch(scans, tfs)
    names = (C.c_char * (64 * 64))()
    ms = (C.c_double * 64)()
    calls = (C.c_uint64 * 64)()
    n = lib.profile_read(det.h, names, ms, calls, 64)
    lib.profile_enable(det.h, 0)
    kernels = {}
    for i in range(n):
        nm = names[64 * i : 64 * i + 64].split(b"\0", 1)[0].decode()
        kernels[nm] = {"avg_us": 1e3 * ms[i] / max(calls[i], 1), "launches": int(calls[i])}
    # algorithmic bytes per launch: every launch covers F frames
    M = det.n_voxels
    alg = {
        # batched fast path (kernels_frame.h)
        "k_bbox": 12.0 * n_pts * F,          # bounding box: one read of the xyz columns
        # round 5: the frame kernel reads the xyz columns itself (one pass: 12*N) ...
        "k_frame_lds_full": (12.0 * n_pts + 40.0 * V) * F,  # ... weighted cloud out 16*V, clustering in 16*V, labels 4*V, member list 4*V
        # close first (round 4): the weighted cloud out 16*V, one bit of the dilated map image per voxel, and the clustering's
        # 16 + 4 + 4 bytes only for the voxels of the far clusters - what this kernel HAS to move, not the contract's 40*V
        "k_frame_lds_far": (12.0 * n_pts + 16.0 * V + V / 8.0 + 24.0 * V_far) * F,
        # general path (single scans, fallbacks)
        "k_setbits": 12.0 * n_pts * F,
        "k_key": 12.0 * n_pts * F,
        "k_slab": 4.0 * V * F,
        "k_slab_emit": 28.0 * V * F,
        "k_flatten<2>": 8.0 * V * F,
        "k_count": 12.0 * n_pts * F,
        "k_emit": 20.0 * V * F,
        "k_union<2>": 20.0 * V * F,
        "k_flatten<0>": 8.0 * V * F,
        "k_flatten<1>": 8.0 * V * F,
        "k_brick_set": 4.0 * V * F,
        "k_brick_union<1>": 20.0 * V * F,
        "k_brick_conn": 16.0 * V * F,
        "k_brick_link_tr": 4.0 * V * F,
        "k_brick_link": 4.0 * V * F,
        "k_brick_root": 4.0 * V * F,
        "k_closefar": 4.0 * V * F,
        "k_finalize": 12.0 * V * F,
        "k_mapbits": 4.0 * M + M / 8.0,
    }
    for nm, k in kernels.items():
        if nm in alg and k["avg_us"] > 0:
            k["alg_bytes"] = alg[nm]
            k["GBps"] = alg[nm] / (k["avg_us"] * 1e-6) / 1e9
    # the voxelize+cluster path of north_star: K1-K7 (either clustering family)
    path_prefixes = ("k_init_hdr", "k_bbox", "k_grid", "k_setbits", "k_key", "k_frame_lds", "k_slab", "k_scan_a", "k_scan_b", "k_emit", "k_count", "k_union", "k_flatten", "k_brick_")
    path = [k for k in kernels if k.startswith(path_prefixes)]
    path_us = sum(kernels[p]["avg_us"] for p in path)
    dom = max(path, key=lambda p: kernels[p]["avg_us"])
    dk = kernels[dom]
    # HBM traffic of the dominant kernel from the PMC passes of this same command (tools/run_profiles.sh ->
    # profiles/rNN_traffic.json, corrected per MI355X_MICROARCH.md).  The file records the hash of the kernel sources it was
    # measured on: a number from other sources would be stale and is not reported.
    traffic, traffic_note = None, "no PMC summary for these kernel sources (run tools/run_profiles.sh)"
    for tr_file in sorted((ROOT / "profiles").glob("r*_traffic.json"), reverse=True):  # the newest round's summary first
        tr = json.loads(tr_file.read_text())
        if tr.get("kernel_source_sha") == kernel_source_sha():
            # (the summary names a kernel as the profiler does: the launch aliases of the frame kernel are its two instantiations)
            alias = {"k_frame_lds_far": "k_frame_lds<1>", "k_frame_lds_full": "k_frame_lds<0>"}.get(dom, dom)
            key = alias if alias in tr else dom if dom in tr else dom.split("<")[0]  # (the summary drops non-numeric template arguments)
            if key not in tr and dom.startswith("k_frame_lds") and "k_frame_lds" in tr:
                key = "k_frame_lds"  # (round 5: two template arguments, <1, true> - the summary keeps the bare name; the profiled command launches the far instantiation only)
            if key in tr:
                traffic = tr[key]["hbm_bytes_per_launch_corrected"]
                traffic_note = f"profiles/{tr_file.name} (FETCH_SIZE x2 + WRITE_SIZE, rocprofv3 --pmc, same command, same kernel sources)"
            break
        traffic_note = f"profiles/{tr_file.name} was measured on other kernel sources: stale, not reported"
    roofline = {
        "bound": "hbm",
        "kernel": dom,
        "achieved": dk.get("GBps", 0.0),
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": dk.get("GBps", 0.0) / HBM_PEAK_GBS,
        "traffic": traffic,
        "traffic_source": traffic_note,
        "alg_bytes_per_launch": dk.get("alg_bytes"),
        "avg_launch_us": dk["avg_us"],
        "path": {
            "what": "voxelize+cluster path, B = (12*N + 40*V) per frame (SURVEY 8d) over the summed kernel time of one batch",
            "alg_bytes_per_batch": (12.0 * n_pts + 40.0 * V) * F,
            "device_us_per_batch": path_us,
            "GBps": (12.0 * n_pts + 40.0 * V) * F / (path_us * 1e-6) / 1e9 if path_us else 0.0,
            "frac": (12.0 * n_pts + 40.0 * V) * F / (path_us * 1e-6) / 1e9 / HBM_PEAK_GBS if path_us else 0.0,
        },
    }
    # The contract's B prices a clustering of ALL voxels.  Read-only batches cluster close first: the background is never
    # clustered at all, so the bytes this path must really move are fewer - both are reported, the dominant kernel's `frac`
    # above is priced on the latter (VERDICT r3 "honesty rule").
    moved = (12.0 * n_pts + 16.0 * V + V / 8.0 + 24.0 * V_far) * F
    roofline["path_moved"] = {
        "what": "bytes the close-first path has to move: 12*N in, 16*V weighted cloud out, V/8 close bits, 24 B per voxel of a far cluster (V_far)",
        "voxels_in_far_clusters_per_frame": V_far,
        "alg_bytes_per_batch": moved,
        "device_us_per_batch": path_us,
        "GBps": moved / (path_us * 1e-6) / 1e9 if path_us else 0.0,
        "frac": moved / (path_us * 1e-6) / 1e9 / HBM_PEAK_GBS if path_us else 0.0,
    }
    return roofline, kernels


def kernel_source_sha():
    """hash of the kernel sources: ties a PMC traffic summary to the code it was measured on.  Comments and white space do not
    count (a corrected comment is not a new kernel): // and /* */ comments are cut and runs of white space collapsed before
    hashing - crude on purpose (a "//" inside a string literal is cut too): the hash only has to change when the code does.
    The files with the device code of the per-scan path count (kernels_*.h, the structures they share in common.h, the device
    eigen-solver); the host driver (vofod_hip.hip, driver_aux.h, collective.h, thread_pool.h, host_tail.h) does not: a change of
    the sepclusters role's allocation policy is not a new frame kernel."""
    import hashlib
    import re

    hsh = hashlib.sha1()
    csrc = ROOT / "vofod_amd" / "csrc"
    for f in sorted(csrc.glob("kernels_*.h")) + [csrc / "common.h", csrc / "eigsolve3.h"]:
        if not f.exists():
            continue
        text = f.read_text(errors="replace")
        text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
        text = re.sub(r"//[^\n]*", " ", text)
        hsh.update(f.name.encode() + b"\0" + " ".join(text.split()).encode())
    return "c2:" + hsh.hexdigest()


def cpu_baseline(args, gpu_det, host_scans):
    """The CPU oracle ("port": restatement of the reference algorithm, 1 thread as pointcloud_threads: 1) timed on
    a bounded sample of the same workload, starting from the very map the GPU run used."""
    from vofod_amd import capi

    so = ROOT / "oracle" / "libvofod_oracle.so"
    olib = capi.Library(so, "vofod_oracle_")
    det = build_detector(olib, args.sensor, args.voxel_size, 1, 0)
    st = gpu_det.status()
    if st.background_pts_sufficient and st.sure_background_sufficient:
        det.load_apriori(np.zeros((0, 3), dtype=np.float32))  # sets both latches, touches no voxel
    det.write_map(capi.MAP_VOXELS, gpu_det.read_map(capi.MAP_VOXELS))
    n = args.cpu_baseline_scans
    t0 = time.perf_counter()
    for i in range(n):
        s = host_scans[i % len(host_scans)]
        det.process_scan(s.scan, s.tf, flags=capi.SCAN_NO_MAP_UPDATE)
    dt = time.perf_counter() - t0
    # per-stage host wall clock of the same oracle under the reference's ScopeTimer checkpoint names
    # (vofod_nodelet.cpp:929-964), mean over a small sample (the debug call also copies the clouds out)
    names = ["filtering", "clusterization", "close X far", "vmap update", "classification"]
    acc = np.zeros(len(names))
    ns = min(16, len(host_scans))
    for i in range(ns):
        _, dbg = det.process_scan(host_scans[i].scan, host_scans[i].tf, flags=capi.SCAN_NO_MAP_UPDATE, debug=True)
        acc += np.array(dbg["stage_ms"][: len(names)])
    stage_ms = {nm: float(v / ns) for nm, v in zip(names, acc)}
    return {
        "stage_ms": stage_ms,
        "stage_ms_note": f"mean over {ns} scans, mrs_lib::ScopeTimer checkpoint names of processMsg (vofod_nodelet.cpp:929-964); read-only map: 'vmap update' is empty",
        "value": n / dt,
        "unit": "frames/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{n} process_scan calls over the benchmark's {len(host_scans)} OS1-128 scans through the C++ CPU oracle (restatement of the reference "
        f"algorithm, not the PCL build; read-only map as in the batched mode) on the GPU run's own warmed map, {dt:.1f} s of CPU work, host has {os.cpu_count()} cores",
    }


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""How many voxels of a bench frame are *far* (no background voxel within hasCloseTo's stencil) on the warmed map?
CPU only (oracle): sizes the close-first clustering of k_frame_lds."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from vofod_amd import capi, synth
from vofod_amd.detector import VoFOD, default_params

def main(n_frames=8, warm=96, sensor="os1-128", vs=0.25):
    lib = capi.Library(str(ROOT / "oracle" / "libvofod_oracle.so"), prefix="vofod_oracle_")
    h, w, vfov_deg, _ = synth.SENSORS[sensor]
    sp, dp = default_params(lib)
    sp.voxel_size = vs
    sp.sensor_hrays, sp.sensor_vrays = w, h
    sp.sensor_vfov = np.float32(np.deg2rad(vfov_deg))
    sp.max_batch_frames = n_frames
    det = VoFOD(lib, sp, dp)
    scene = synth.bench_scene()
    t0 = time.time()
    synth.warm_map(det, scene, sensor, warm)
    print(f"warmed in {time.time()-t0:.1f}s", flush=True)
    np.save(ROOT / "gpurun_out" / f"warm_map_{sensor}_{vs}_{warm}.npy", det.read_map(capi.MAP_VOXELS))
    import ctypes as C
    hct = lib.extra("vofod_oracle_map_has_close_to", C.c_int, [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float])
    frames = synth.bench_frames(scene, sensor, n_frames, 0)
    thr = float(dp.voxel_map__thresholds__new_obstacles)
    md = float(dp.ground_points_max_distance)
    for f, s in enumerate(frames):
        dets, per, dbg = det.process_batch([s.scan], np.stack([s.tf]), debug=True)
        d = dbg[0]
        wpts = d["weighted"]
        V = len(wpts)
        close = np.array([hct(det.h, float(p["x"]), float(p["y"]), float(p["z"]), md, thr) for p in wpts], dtype=bool)
        cl = d["clusters"]
        far_cl = cl[cl["is_close"] == 0]
        print(f"frame {f}: V={V} far_voxels={int((~close).sum())} clusters={len(cl)} far_clusters={len(far_cl)} voxels_in_far_clusters={int(far_cl['n_points'].sum())} dets={len(dets)}", flush=True)

if __name__ == "__main__":
    main(*[int(a) for a in sys.argv[1:3]])

"""Seeded synthetic Ouster-shaped scans.

Sensor model = the simulated LUT of vofod_nodelet.cpp:374-420:
``yaw_c = c*2pi/(w-1)``, ``pitch_r = r*vfov/(h-1) - vfov/2``,
``dir = (cos p cos y, cos p sin y, sin p)``, index ``r*w + c``, beam offsets 0.
Scene: ground plane z=0 inside the operation area, axis-aligned boxes (buildings/
trees) and a few small floating boxes (the MAV-like targets).  Range = exact ray
distance + N(0, 0.02 m), quantised to uint32 millimetres; rays without a return (or
beyond the sensor's max range) carry range 0 and xyz (0,0,0) as ouster_ros does.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from .detector import ScanData

SENSORS = {
    # name: (vrays, hrays, vfov_deg, max_range_m)
    "os1-16": (16, 1024, 33.2, 120.0),
    "os1-128": (128, 1024, 45.0, 120.0),      # config/sensors/os1-128.yaml
    "os0-128": (128, 1024, 90.0, 50.0),       # config/sensors/os0-128.yaml
    "os2-128x2048": (128, 2048, 22.5, 240.0),
}


def sim_lut(w: int, h: int, vfov: float) -> np.ndarray:
    """[h*w, 3] float32 unit directions (vofod_nodelet.cpp:374-420)."""
    yaw = np.arange(w, dtype=np.float64) * (2.0 * np.pi / (w - 1))
    pitch = np.arange(h, dtype=np.float64) * (np.float64(np.float32(vfov)) / (h - 1)) - np.float64(np.float32(vfov)) / 2.0
    cp, sp = np.cos(pitch)[:, None], np.sin(pitch)[:, None]
    d = np.stack([cp * np.cos(yaw)[None, :], cp * np.sin(yaw)[None, :], np.broadcast_to(sp, (h, w))], axis=-1)
    return d.reshape(-1, 3).astype(np.float32)


@dataclass
class Scene:
    ground_rect: tuple  # (xmin, xmax, ymin, ymax) of the z=0 ground sheet
    boxes: np.ndarray   # [B, 6] xmin,ymin,zmin,xmax,ymax,zmax
    n_static: int       # the first n_static boxes are buildings, the rest floating targets
    seed: int = 0


def make_scene(seed: int = 0, oparea_offset=(40.0, 20.0, -1.25), oparea_size=(120.0, 100.0, 25.0), n_targets: int | None = None,
               sensor_xy=(0.0, 0.0)) -> Scene:
    rng = np.random.default_rng(seed)
    cx, cy, zb = oparea_offset
    sx, sy, sz = oparea_size
    x0, x1, y0, y1 = cx - sx / 2, cx + sx / 2, cy - sy / 2, cy + sy / 2
    boxes = []
    n_static = int(rng.integers(6, 11))
    for _ in range(n_static):
        w, d, hgt = rng.uniform(2.0, 15.0), rng.uniform(2.0, 15.0), rng.uniform(2.0, 15.0)
        while True:
            bx, by = rng.uniform(x0 + 1, x1 - w - 1), rng.uniform(y0 + 1, y1 - d - 1)
            # keep the sensor's immediate surroundings free
            if not (bx - 4 < sensor_xy[0] < bx + w + 4 and by - 4 < sensor_xy[1] < by + d + 4):
                break
        boxes.append([bx, by, 0.0, bx + w, by + d, hgt])
    nt = int(rng.integers(1, 4)) if n_targets is None else n_targets
    for _ in range(nt):
        s = rng.uniform(0.4, 0.8)
        r, a = rng.uniform(5.0, 40.0), rng.uniform(0, 2 * np.pi)
        tx, ty = sensor_xy[0] + r * np.cos(a), sensor_xy[1] + r * np.sin(a)
        tx = float(np.clip(tx, x0 + 2, x1 - 2))
        ty = float(np.clip(ty, y0 + 2, y1 - 2))
        tz = rng.uniform(3.0, min(zb + sz - 2.0, 18.0))
        boxes.append([tx - s / 2, ty - s / 2, tz - s / 2, tx + s / 2, ty + s / 2, tz + s / 2])
    return Scene((x0, x1, y0, y1), np.asarray(boxes, dtype=np.float64), n_static, seed)


def make_pose(seed: int = 0, xy=(0.0, 0.0)) -> np.ndarray:
    """Row-major 3x4 [R|t] float32: t=(x,y,4..10), random yaw, roll/pitch within +-5 deg."""
    rng = np.random.default_rng(seed + 7919)
    z = rng.uniform(4.0, 10.0)
    yaw = rng.uniform(-np.pi, np.pi)
    roll, pitch = np.deg2rad(rng.uniform(-5, 5, size=2))
    cy_, sy_ = np.cos(yaw), np.sin(yaw)
    cp_, sp_ = np.cos(pitch), np.sin(pitch)
    cr_, sr_ = np.cos(roll), np.sin(roll)
    Rz = np.array([[cy_, -sy_, 0], [sy_, cy_, 0], [0, 0, 1]])
    Ry = np.array([[cp_, 0, sp_], [0, 1, 0], [-sp_, 0, cp_]])
    Rx = np.array([[1, 0, 0], [0, cr_, -sr_], [0, sr_, cr_]])
    R = Rz @ Ry @ Rx
    tf = np.zeros((3, 4), dtype=np.float64)
    tf[:, :3] = R
    tf[:, 3] = (xy[0], xy[1], z)
    return tf.astype(np.float32)


@dataclass
class SynthScan:
    scan: ScanData
    tf: np.ndarray
    x: np.ndarray
    y: np.ndarray
    z: np.ndarray
    intensity: np.ndarray
    range: np.ndarray
    keep: list = field(default_factory=list)


def make_scan(scene: Scene, tf: np.ndarray, sensor: str = "os1-128", seed: int = 0, noise_sigma: float = 0.02) -> SynthScan:
    h, w, vfov_deg, max_range = SENSORS[sensor]
    vfov = np.float32(np.deg2rad(vfov_deg))
    rng = np.random.default_rng(seed + 104729)
    dirs_s = sim_lut(w, h, float(vfov)).astype(np.float64)
    R = tf[:, :3].astype(np.float64)
    o = tf[:, 3].astype(np.float64)
    d = dirs_s @ R.T  # world directions
    n = d.shape[0]
    best = np.full(n, np.inf)
    # ground sheet z = 0
    with np.errstate(divide="ignore", invalid="ignore"):
        tg = -o[2] / d[:, 2]
        gx, gy = o[0] + tg * d[:, 0], o[1] + tg * d[:, 1]
        x0, x1, y0, y1 = scene.ground_rect
        ok = (tg > 0) & (gx >= x0) & (gx <= x1) & (gy >= y0) & (gy <= y1)
        best = np.where(ok, tg, best)
        # boxes: slab test
        inv = 1.0 / d
        for b in scene.boxes:
            t1 = (b[:3] - o) * inv
            t2 = (b[3:] - o) * inv
            tmin = np.nanmax(np.minimum(t1, t2), axis=1)
            tmax = np.nanmin(np.maximum(t1, t2), axis=1)
            hit = (tmax >= np.maximum(tmin, 0.0)) & (tmin > 0)
            best = np.where(hit & (tmin < best), tmin, best)
    rng_m = best + rng.normal(0.0, noise_sigma, size=n)
    valid = np.isfinite(best) & (rng_m > 0.3) & (rng_m <= max_range)
    range_mm = np.where(valid, np.round(rng_m * 1000.0), 0).astype(np.uint32)
    dirs32 = dirs_s.astype(np.float32)
    r32 = (range_mm.astype(np.float32) * np.float32(0.001)).astype(np.float32)
    x = (dirs32[:, 0] * r32).astype(np.float32)
    y = (dirs32[:, 1] * r32).astype(np.float32)
    z = (dirs32[:, 2] * r32).astype(np.float32)
    intensity = rng.uniform(0.0, 1000.0, size=n).astype(np.float32)
    scan = ScanData(x=x, y=y, z=z, width=w, height=h, intensity=intensity, range=range_mm, stride_bytes=4)
    return SynthScan(scan=scan, tf=tf.astype(np.float32), x=x, y=y, z=z, intensity=intensity, range=range_mm)


def apriori_points(scene: Scene, voxel_size: float, n_voxels: int | None = None, seed: int = 0, solid_ground_to: float | None = None) -> np.ndarray:
    """World-frame points on the ground sheet and on the static boxes' shells, one per
    voxel-sized cell (stand-in for the downsampled static cloud of vofod_nodelet.cpp:332-341).
    `solid_ground_to`: also fill the ground below z = 0 down to that height, one point per voxel layer (a surveyed
    terrain volume: what it takes to reach the 1 M apriori voxels of BASELINE.json configs[2] at 0.25 m)."""
    rng = np.random.default_rng(seed + 15485863)
    vs = voxel_size
    x0, x1, y0, y1 = scene.ground_rect
    gx, gy = np.meshgrid(np.arange(x0 + vs / 2, x1, vs), np.arange(y0 + vs / 2, y1, vs), indexing="ij")
    pts = [np.stack([gx.ravel(), gy.ravel(), np.full(gx.size, 0.0 + vs * 0.25)], axis=1)]
    if solid_ground_to is not None:
        zl = vs * 0.25 - vs
        while zl > solid_ground_to:
            pts.append(np.stack([gx.ravel(), gy.ravel(), np.full(gx.size, zl)], axis=1))
            zl -= vs
    for b in scene.boxes[: scene.n_static]:
        xs = np.arange(b[0] + vs / 2, b[3], vs)
        ys = np.arange(b[1] + vs / 2, b[4], vs)
        zs = np.arange(b[2] + vs / 2, b[5], vs)
        for xv in (b[0] + 1e-3, b[3] - 1e-3):
            yy, zz = np.meshgrid(ys, zs, indexing="ij")
            pts.append(np.stack([np.full(yy.size, xv), yy.ravel(), zz.ravel()], axis=1))
        for yv in (b[1] + 1e-3, b[4] - 1e-3):
            xx, zz = np.meshgrid(xs, zs, indexing="ij")
            pts.append(np.stack([xx.ravel(), np.full(xx.size, yv), zz.ravel()], axis=1))
        xx, yy = np.meshgrid(xs, ys, indexing="ij")
        pts.append(np.stack([xx.ravel(), yy.ravel(), np.full(xx.size, b[5] - 1e-3)], axis=1))
    p = np.concatenate(pts, axis=0)
    if n_voxels is not None and p.shape[0] > n_voxels:
        p = p[rng.choice(p.shape[0], size=n_voxels, replace=False)]
    return p.astype(np.float32)


# ---------------------------------------------------------------------------- scenarios

def seed_ground(det, xy=(0.0, 0.0), radius: float = 30.0, value: float = -185.0):
    """Stand-in for the range-finder ground seeding of processMsg(Range) (vofod_nodelet.cpp:581-613,
    out of scope here): marks the ground voxels around the sensor as background so that the first
    scans have something to be 'close' to (a hovering OS1 does not see the ground right below it).  Works on any implementation of the C-ABI."""
    from . import capi

    m = det.read_map(capi.MAP_VOXELS)
    sx, sy, sz = det.map_size
    ox, oy, oz = det.map_offset
    vs = float(det.sp.voxel_size)
    iz = int(np.floor((0.0 - oz) / vs))
    r = int(np.ceil(radius / vs))
    cx, cy = int(np.floor((xy[0] - ox) / vs)), int(np.floor((xy[1] - oy) / vs))
    x0, x1 = max(cx - r, 0), min(cx + r + 1, sx)
    y0, y1 = max(cy - r, 0), min(cy + r + 1, sy)
    if 0 <= iz < sz:
        m[iz, y0:y1, x0:x1] = value
    det.write_map(capi.MAP_VOXELS, m)


def scan_sequence(scene: Scene, sensor: str, n: int, seed0: int = 0, xy=(0.0, 0.0)):
    """n scans of one scene from slowly varying poses (a hovering MAV)."""
    out = []
    for k in range(n):
        tf = make_pose(seed0 + k, xy=xy)
        out.append(make_scan(scene, tf, sensor, seed=seed0 + k))
    return out


# ---------------------------------------------------------------------------- bench workload

BENCH_SCENE_SEED = 0
BENCH_WARM_SEED0 = 1000


def bench_scene() -> Scene:
    """the scene bench.py measures on (BASELINE.json configs[1]/[3]): seed 0, three floating targets"""
    return make_scene(BENCH_SCENE_SEED, n_targets=3)


def bench_frames(scene: Scene, sensor: str, n: int, rank: int = 0):
    """the n independent scans of one rank's batch in bench.py: independent poses and noise seeds 10 000*rank + f"""
    return [make_scan(scene, make_pose(10_000 * rank + f), sensor, seed=10_000 * rank + f) for f in range(n)]


def warm_map(det, scene: Scene, sensor: str, n_scans: int, seed0: int = BENCH_WARM_SEED0):
    """config 2 of SURVEY 8d: map pre-warmed by n scans with the raycast and sepclusters roles interleaved.
    Works on any implementation of the C-ABI (bench.py warms the HIP detector, the tests hand the map to the oracle)."""
    from . import capi

    seed_ground(det)
    for k, s in enumerate(scan_sequence(scene, sensor, n_scans, seed0=seed0)):
        det.process_scan(s.scan, s.tf, flags=capi.SCAN_AUTO_RAYCAST)
        if k % 2 == 1:
            st, sure = det.sepclusters_begin(allow=(capi.ERR_EMPTY,))
            if st == capi.OK and sure:
                det.sepclusters_finish()
    if det.status().raycast_pending:
        det.raycast_finish(allow=(capi.ERR_RAYCAST_NO_DETECTION, capi.ERR_RAYCAST_EMPTY))

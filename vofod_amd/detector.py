"""Host-side mirror of the vofod::VoFOD nodelet interface for the per-scan hot path.

`VoFOD` wraps one `vofod_handle` of a `capi.Library` and exposes the same call
order the nodelet uses (vofod_nodelet.cpp:882-989): `process_scan` (= the body of
processMsg), `raycast_begin/finish` (= raycast_cloud :1397), `sepclusters_begin/
finish` (= updateSeparatedBGClusters :1126), `reset` (:1610), `load_apriori`
(:339-345).  Parameter names are those of config/detection_params.yaml.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Sequence

import numpy as np

from . import capi


class VofodError(RuntimeError):
    def __init__(self, status: int, what: str, detail: str = ""):
        super().__init__(f"{what} failed with status {status}" + (f": {detail}" if detail else ""))
        self.status = status


@dataclass
class ScanData:
    """One organised scan.  Arrays are float32/uint32 of w*h elements (host numpy),
    or integer device addresses with `memspace=capi.MEM_DEVICE`."""

    x: object
    y: object
    z: object
    width: int
    height: int
    intensity: object = None
    range: object = None
    stride_bytes: int = 4
    memspace: int = capi.MEM_HOST
    stamp: float = 0.0

    def as_c(self) -> capi.Scan:
        def p(a):
            if a is None:
                return None
            if isinstance(a, np.ndarray):
                return a.ctypes.data_as(C.c_void_p)
            return C.c_void_p(int(a))

        s = capi.Scan()
        s.x, s.y, s.z = p(self.x), p(self.y), p(self.z)
        s.intensity, s.range = p(self.intensity), p(self.range)
        s.stride_bytes = self.stride_bytes
        s.width, s.height = self.width, self.height
        s.memspace = self.memspace
        s.stamp = self.stamp
        return s


def default_params(lib: capi.Library) -> tuple[capi.StaticParams, capi.DynParams]:
    sp, dp = capi.StaticParams(), capi.DynParams()
    lib.default_params(C.byref(sp), C.byref(dp))
    return sp, dp


class VoFOD:
    def __init__(self, lib: capi.Library, sp: capi.StaticParams | None = None, dp: capi.DynParams | None = None,
                 lut_directions: np.ndarray | None = None, lut_offsets: np.ndarray | None = None, mask: np.ndarray | None = None):
        self.lib = lib
        dsp, ddp = default_params(lib)
        self.sp = sp if sp is not None else dsp
        self.dp = dp if dp is not None else ddp
        self._keep = []
        n = self.sp.sensor_hrays * self.sp.sensor_vrays
        for name, arr, dt, cnt in (("lut_directions", lut_directions, np.float32, 3 * n), ("lut_offsets", lut_offsets, np.float32, 3 * n), ("mask", mask, np.uint8, n)):
            if arr is not None:
                a = np.ascontiguousarray(arr, dtype=dt).reshape(-1)
                assert a.size == cnt, f"{name}: expected {cnt} elements, got {a.size}"
                self._keep.append(a)
                setattr(self.sp, name, a.ctypes.data_as(C.c_void_p))
        self._scan_arrays = {}
        self._pending = {}
        self._collect_bufs = {}
        self.h = C.c_void_p()
        st = lib.create(C.byref(self.sp), C.byref(self.dp), C.byref(self.h))
        if st != capi.OK:
            raise VofodError(st, "vofod_create")
        info = self.status()
        self.map_size = tuple(info.map_size)  # (sx, sy, sz)
        self.map_offset = tuple(info.map_offset)
        self.n_voxels = int(np.prod(self.map_size))

    # ------------------------------------------------------------ lifecycle
    def close(self):
        if self.h:
            self.lib.destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st: int, what: str, allow: Sequence[int] = ()):
        if st != capi.OK and st not in allow:
            msg = self.lib.last_error_string(self.h)
            raise VofodError(st, what, msg.decode() if msg else "")
        return st

    def reset(self):
        self._check(self.lib.reset(self.h), "vofod_reset")

    def set_dynamic_params(self, **kv):
        for k, v in kv.items():
            if not hasattr(self.dp, k):
                raise KeyError(k)
            setattr(self.dp, k, v)
        self._check(self.lib.set_dynamic_params(self.h, C.byref(self.dp)), "vofod_set_dynamic_params")

    def status(self) -> capi.StatusInfo:
        s = capi.StatusInfo()
        self._check(self.lib.get_status(self.h, C.byref(s)), "vofod_get_status")
        return s

    def load_apriori(self, xyz: np.ndarray):
        a = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        self._check(self.lib.load_apriori(self.h, capi.ptr(a), a.shape[0]), "vofod_load_apriori")

    def ingest_apriori(self, filename: str, tf_xyz=(0.0, 0.0, 0.0), yaw_deg: float = 0.0, sim_correction=(0.0, 0.0, 0.0)):
        """initialize_apriori_map from a .pts/.xyz file (apriori_map/tf/* of sim.yaml); returns (points loaded, voxels set)"""
        t = np.ascontiguousarray(tf_xyz, dtype=np.float32)
        c = np.ascontiguousarray(sim_correction, dtype=np.float32)
        nl, nv = C.c_size_t(0), C.c_size_t(0)
        self._check(self.lib.ingest_apriori(self.h, filename.encode(), capi.ptr(t), float(yaw_deg), capi.ptr(c), C.byref(nl), C.byref(nv)), "vofod_ingest_apriori")
        return nl.value, nv.value

    def update_ground(self, range_m: float, tf, min_range: float = 0.0, max_range: float = 100.0, allow=()):
        """height range-finder message (vofod_nodelet.cpp:581-613); returns the status (MAP_RANGE may be allowed)"""
        t = np.ascontiguousarray(tf, dtype=np.float32).reshape(12)
        st = self.lib.update_ground(self.h, float(range_m), float(min_range), float(max_range), capi.ptr(t))
        if st != capi.OK and st not in allow:
            self._check(st, "vofod_update_ground")
        return st

    def read_map(self, which: int = capi.MAP_VOXELS) -> np.ndarray:
        """Returns the map as [sz, sy, sx] (x fastest: voxel_map.cpp:81)."""
        out = np.empty(self.n_voxels, dtype=np.float32)
        self._check(self.lib.read_map(self.h, which, capi.ptr(out), out.size), "vofod_read_map")
        sx, sy, sz = self.map_size
        return out.reshape(sz, sy, sx)

    def voxels_as_pc(self, threshold: float, greater_than: bool = True, which: int = capi.MAP_VOXELS) -> np.ndarray:
        """VoxelMap::voxelsAsPC (voxel_map.cpp:157-183): [n, 4] float32 world centres + map value, x outer / y / z inner"""
        n = C.c_size_t(0)
        st = self.lib.voxels_as_pc(self.h, which, float(threshold), int(bool(greater_than)), None, 0, C.byref(n))
        if st not in (capi.OK, capi.ERR_CAPACITY):
            self._check(st, "vofod_voxels_as_pc")
        out = np.zeros((n.value, 4), dtype=np.float32)
        if n.value:
            self._check(self.lib.voxels_as_pc(self.h, which, float(threshold), int(bool(greater_than)), capi.ptr(out), n.value, C.byref(n)), "vofod_voxels_as_pc")
        return out

    def write_map(self, which: int, arr: np.ndarray):
        a = np.ascontiguousarray(arr, dtype=np.float32).reshape(-1)
        self._check(self.lib.write_map(self.h, which, capi.ptr(a), a.size), "vofod_write_map")

    # ------------------------------------------------------------- hot path
    def _mk_dbg(self, n_points: int, clusters_cap: int | None = None):
        """debug buffers of one frame; `clusters_cap` bounds the cluster table (default: one per point, the worst case)"""
        n_cl = n_points if clusters_cap is None else clusters_cap
        w = np.zeros(n_points, dtype=capi.POINT_XYZR)
        lab = np.zeros(n_points, dtype=np.uint32)
        cl = np.zeros(n_cl, dtype=capi.CLUSTER_INFO)
        d = capi.ScanDebug()
        d.weighted, d.labels, d.weighted_cap = capi.ptr(w), capi.ptr(lab), n_points
        d.clusters, d.clusters_cap = capi.ptr(cl), n_cl
        return d, (w, lab, cl)

    @staticmethod
    def _dbg_dict(d: capi.ScanDebug, bufs):
        w, lab, cl = bufs
        return {
            "weighted": w[: d.n_weighted].copy(),
            "labels": lab[: d.n_weighted].copy(),
            "clusters": cl[: d.n_clusters].copy(),
            "n_input_after_crop": int(d.n_input_after_crop),
            "n_bg_voxels": int(d.n_bg_voxels),
            "background_pts_sufficient": bool(d.background_pts_sufficient),
            "sure_background_sufficient": bool(d.sure_background_sufficient),
            "stage_ms": list(d.stage_ms),
        }

    def process_scan(self, scan: ScanData, tf: np.ndarray, flags: int = capi.SCAN_DEFAULT, debug: bool = False, det_cap: int = 256,
                     allow: Sequence[int] = ()):
        tfa = np.ascontiguousarray(tf, dtype=np.float32).reshape(12)
        dets = np.zeros(det_cap, dtype=capi.DETECTION)
        n_out = C.c_size_t(0)
        cs = scan.as_c()
        dbg, bufs = (self._mk_dbg(scan.width * scan.height) if debug else (None, None))
        st = self.lib.process_scan(self.h, C.byref(cs), capi.ptr(tfa), flags, capi.ptr(dets), det_cap, C.byref(n_out), C.byref(dbg) if debug else None)
        self._check(st, "vofod_process_scan", allow)
        out = dets[: n_out.value].copy()
        return (out, self._dbg_dict(dbg, bufs)) if debug else out

    def process_batch(self, scans: Sequence[ScanData], tfs: np.ndarray, debug: bool = False, det_cap: int = 4096, clusters_cap: int | None = None, far_only: bool = False):
        """`far_only` (with `debug`): the view of the production path of read-only batches, which clusters close first - the
        cluster table lists the far clusters only, labels are capi.LABEL_NONE outside them (include/vofod.h)."""
        n = len(scans)
        arr = (capi.Scan * n)(*[s.as_c() for s in scans])
        tfa = np.ascontiguousarray(tfs, dtype=np.float32).reshape(n, 12)
        dets = np.zeros(det_cap, dtype=capi.DETECTION)
        per = np.zeros(n, dtype=np.uint32)
        n_out = C.c_size_t(0)
        dbgs = None
        bufs = []
        if debug:
            dbgs = (capi.ScanDebug * n)()
            for f in range(n):
                d, b = self._mk_dbg(scans[f].width * scans[f].height, clusters_cap)
                d.far_only = 1 if far_only else 0
                dbgs[f] = d
                bufs.append(b)
        st = self.lib.process_batch(self.h, arr, capi.ptr(tfa), n, capi.ptr(dets), det_cap, capi.ptr(per), C.byref(n_out), dbgs)
        self._check(st, "vofod_process_batch")
        out = dets[: n_out.value].copy()
        if debug:
            return out, per, [self._dbg_dict(dbgs[f], bufs[f]) for f in range(n)]
        return out, per

    def reserve(self, tickets: int):
        """allocate the workspaces of `tickets` batches in flight now instead of inside the first submits"""
        self._check(self.lib.reserve(self.h, int(tickets)), "vofod_reserve")

    def batch_submit(self, scans: Sequence[ScanData], tfs: np.ndarray) -> int:
        """Enqueue a batch (read-only map); returns the ticket for `batch_collect`.  At most eight in flight."""
        n = len(scans)
        key = id(scans)
        cached = self._scan_arrays.get(key)
        if cached is None or cached[0] != n:
            cached = (n, (capi.Scan * n)(*[s.as_c() for s in scans]), scans)  # keeps `scans` alive: the key is its id
            self._scan_arrays[key] = cached
        tfa = np.ascontiguousarray(tfs, dtype=np.float32).reshape(n, 12)
        ticket = C.c_int(-1)
        self._check(self.lib.batch_submit(self.h, cached[1], capi.ptr(tfa), n, C.byref(ticket)), "vofod_batch_submit")
        self._pending[ticket.value] = n
        return ticket.value

    def batch_collect(self, ticket: int, det_cap: int = 4096):
        n = self._pending.pop(ticket)
        for attempt in (0, 1):
            buf = self._collect_bufs.get((det_cap, n))  # reused between calls: the results are copied out below
            if buf is None:
                buf = (np.zeros(det_cap, dtype=capi.DETECTION), np.zeros(n, dtype=np.uint32))
                self._collect_bufs[(det_cap, n)] = buf
            dets, per = buf
            n_out = C.c_size_t(0)
            st = self.lib.batch_collect(self.h, ticket, capi.ptr(dets), det_cap, capi.ptr(per), C.byref(n_out))
            if st == capi.ERR_CAPACITY and attempt == 0 and n_out.value > det_cap:
                # device-tail batches stay pending when the array was too small (include/vofod.h): come back with the
                # size the library asked for.  (A batch that went through the host tail is consumed: the retry then
                # reports NOT_PENDING and the capacity error is raised.)
                msg = self.lib.last_error_string(self.h)
                det_cap = int(n_out.value)
                continue
            if st == capi.ERR_NOT_PENDING and attempt == 1:
                raise VofodError(capi.ERR_CAPACITY, "vofod_batch_collect", f"more than det_cap detections ({msg.decode() if msg else ''})")
            self._check(st, "vofod_batch_collect")
            return dets[: n_out.value].copy(), per.copy()

    def raycast_begin(self, scan: ScanData, tf: np.ndarray, allow: Sequence[int] = ()):
        tfa = np.ascontiguousarray(tf, dtype=np.float32).reshape(12)
        cs = scan.as_c()
        return self._check(self.lib.raycast_begin(self.h, C.byref(cs), capi.ptr(tfa)), "vofod_raycast_begin", allow)

    def raycast_finish(self, allow: Sequence[int] = ()):
        return self._check(self.lib.raycast_finish(self.h), "vofod_raycast_finish", allow)

    def sepclusters_begin(self, allow: Sequence[int] = ()):
        sure = C.c_int(0)
        st = self._check(self.lib.sepclusters_begin(self.h, C.byref(sure)), "vofod_sepclusters_begin", allow)
        return st, bool(sure.value)

    def sepclusters_finish(self, allow: Sequence[int] = ()):
        return self._check(self.lib.sepclusters_finish(self.h), "vofod_sepclusters_finish", allow)

    # ------------------------------------------------- stateless L4 helpers
    def voxel_grid_weighted(self, x, y, z, leaf: float, align_center=None):
        return voxel_grid_weighted(self.lib, x, y, z, leaf, align_center, handle=self.h)

    def voxel_grid_counted(self, x, y, z, intensity, leaf: float, threshold: float):
        return voxel_grid_counted(self.lib, x, y, z, intensity, leaf, threshold, handle=self.h)

    def cluster(self, pts, keys, grid, tolerance: float):
        return cluster(self.lib, pts, keys, grid, tolerance, handle=self.h)


def _view(x, y, z, intensity=None):
    xs = [np.ascontiguousarray(a, dtype=np.float32).reshape(-1) for a in (x, y, z)]
    v = capi.CloudView()
    v.x, v.y, v.z = (capi.ptr(a) for a in xs)
    if intensity is not None:
        it = np.ascontiguousarray(intensity, dtype=np.float32).reshape(-1)
        xs.append(it)
        v.intensity = capi.ptr(it)
    v.stride_bytes = 4
    v.n = xs[0].size
    v.memspace = capi.MEM_HOST
    return v, xs


def voxel_grid_weighted(lib, x, y, z, leaf, align_center=None, handle=None, allow=()):
    v, keep = _view(x, y, z)
    n = max(int(v.n), 1)
    out = np.zeros(n, dtype=capi.POINT_XYZR)
    keys = np.zeros(n, dtype=np.uint32)
    n_out = C.c_size_t(0)
    grid = capi.GridDesc()
    ac = None if align_center is None else np.ascontiguousarray(align_center, dtype=np.float32)
    st = lib.voxel_grid_weighted(handle, C.byref(v), leaf, int(ac is not None), capi.ptr(ac), capi.ptr(out), capi.ptr(keys), n, C.byref(n_out), C.byref(grid))
    if st != capi.OK and st not in allow:
        raise VofodError(st, "vofod_voxel_grid_weighted")
    return out[: n_out.value].copy(), keys[: n_out.value].copy(), grid, st


def voxel_grid_counted(lib, x, y, z, intensity, leaf, threshold, handle=None, allow=()):
    v, keep = _view(x, y, z, intensity)
    n = max(int(v.n), 1)
    out = np.zeros(n, dtype=capi.POINT_XYZR)
    keys = np.zeros(n, dtype=np.uint32)
    n_out = C.c_size_t(0)
    grid = capi.GridDesc()
    st = lib.voxel_grid_counted(handle, C.byref(v), leaf, threshold, capi.ptr(out), capi.ptr(keys), n, C.byref(n_out), C.byref(grid))
    if st != capi.OK and st not in allow:
        raise VofodError(st, "vofod_voxel_grid_counted")
    return out[: n_out.value].copy(), keys[: n_out.value].copy(), grid, st


def cluster(lib, pts, keys, grid, tolerance, handle=None):
    p = np.ascontiguousarray(pts, dtype=capi.POINT_XYZR)
    k = None if keys is None else np.ascontiguousarray(keys, dtype=np.uint32)
    labels = np.zeros(max(p.size, 1), dtype=np.uint32)
    nc = C.c_size_t(0)
    st = lib.cluster(handle, capi.ptr(p), capi.ptr(k), C.byref(grid) if grid is not None else None, p.size, tolerance, capi.ptr(labels), C.byref(nc))
    if st != capi.OK:
        raise VofodError(st, "vofod_cluster")
    return labels[: p.size].copy(), nc.value


def load_cloud(lib, filename: str) -> np.ndarray:
    n = C.c_size_t(0)
    st = lib.load_cloud(filename.encode(), None, 0, C.byref(n))
    if st not in (capi.OK, capi.ERR_CAPACITY):
        raise VofodError(st, "vofod_load_cloud")
    out = np.zeros((n.value, 3), dtype=np.float32)
    if n.value:
        st = lib.load_cloud(filename.encode(), capi.ptr(out), n.value, C.byref(n))
        if st != capi.OK:
            raise VofodError(st, "vofod_load_cloud")
    return out


def sim_lut(lib, w: int, h: int, vfov: float) -> np.ndarray:
    out = np.zeros((h * w, 3), dtype=np.float32)
    st = lib.sim_lut(w, h, vfov, capi.ptr(out))
    if st != capi.OK:
        raise VofodError(st, "vofod_sim_lut")
    return out


def ouster_lut(lib, w: int, h: int, azimuth_deg, altitude_deg, range_unit: float = 0.001, origin_mm: float = 0.0, tf=None):
    """initialize_sensor_lut (vofod_nodelet.cpp:358-372): (directions, offsets), each [h*w, 3] float32"""
    az = np.ascontiguousarray(azimuth_deg, dtype=np.float64)
    alt = np.ascontiguousarray(altitude_deg, dtype=np.float64)
    t = None if tf is None else np.ascontiguousarray(tf, dtype=np.float64).reshape(16)
    d = np.zeros((h * w, 3), dtype=np.float32)
    o = np.zeros((h * w, 3), dtype=np.float32)
    st = lib.ouster_lut(w, h, float(range_unit), float(origin_mm), None if t is None else capi.ptr(t), capi.ptr(az), capi.ptr(alt), capi.ptr(d), capi.ptr(o))
    if st != capi.OK:
        raise RuntimeError(f"vofod_ouster_lut: status {st}")
    return d, o


def mask_layout(lib, image, w: int, h: int, pixel_shift_by_row=None, mangle: bool = True):
    """load_mask (vofod_nodelet.cpp:506-560) after decoding: uint8[w*h]"""
    img = None if image is None else np.ascontiguousarray(image, dtype=np.uint8).reshape(h * w)
    sh = None if pixel_shift_by_row is None else np.ascontiguousarray(pixel_shift_by_row, dtype=np.int32)
    out = np.zeros(h * w, dtype=np.uint8)
    st = lib.mask_layout(None if img is None else capi.ptr(img), w, h, None if sh is None else capi.ptr(sh), int(bool(mangle)), capi.ptr(out))
    if st != capi.OK:
        raise RuntimeError(f"vofod_mask_layout: status {st}")
    return out


def check_sensor_params(lib, scan: ScanData, lut_directions, lut_offsets=None, mask=None):
    """check_sensor_params (vofod_nodelet.cpp:1869-1917): (params_ok, checked) for a host-resident organised scan"""
    d = np.ascontiguousarray(lut_directions, dtype=np.float32).reshape(-1)
    o = None if lut_offsets is None else np.ascontiguousarray(lut_offsets, dtype=np.float32).reshape(-1)
    m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8).reshape(-1)
    cs = scan.as_c()
    checked = C.c_int32(0)
    st = lib.check_sensor_params(C.byref(cs), capi.ptr(d), None if o is None else capi.ptr(o), None if m is None else capi.ptr(m), C.byref(checked))
    if st not in (capi.OK, capi.ERR_SIZE_MISMATCH):
        raise VofodError(st, "vofod_check_sensor_params")
    return st == capi.OK, bool(checked.value)

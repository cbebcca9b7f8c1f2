"""ctypes mirror of include/vofod.h.

`Library(path, prefix)` binds every entry point the header declares under a symbol
prefix, so the same host code drives the product (`libvofod_hip.so`, prefix
``vofod_``) and — from tests/bench only — the CPU oracle (``vofod_oracle_``).
Nothing in this module loads a library by itself.
"""
from __future__ import annotations

import ctypes as C
import re
from pathlib import Path

import numpy as np

HEADER = Path(__file__).resolve().parent.parent / "include" / "vofod.h"

# ------------------------------------------------------------------ status
OK = 0
ERR_INVALID_ARG = 1
ERR_SIZE_MISMATCH = 2
ERR_SENSOR_OUTSIDE_MAP = 3
ERR_INDEX_OVERFLOW = 4
ERR_CAPACITY = 5
ERR_DEVICE = 6
ERR_RAYCAST_NO_DETECTION = 7
ERR_RAYCAST_EMPTY = 8
ERR_PAUSED = 9
ERR_NOT_PENDING = 10
ERR_EMPTY = 11
ERR_MAP_RANGE = 12
ERR_BUSY = 13

MEM_HOST, MEM_DEVICE = 0, 1
MAP_VOXELS, MAP_FLAGS, MAP_RAYCAST = 0, 1, 2
SCAN_DEFAULT, SCAN_NO_MAP_UPDATE, SCAN_AUTO_RAYCAST = 0, 1, 2
CLASS_MAV, CLASS_UNKNOWN, CLASS_INVALID, CLASS_NONE = 0, 1, 2, -1


class StaticParams(C.Structure):
    _fields_ = [
        ("voxel_size", C.c_float),
        ("score_init", C.c_float),
        ("background_sufficient_points_ratio", C.c_float),
        ("oparea_offset", C.c_float * 3),
        ("oparea_size", C.c_float * 3),
        ("exclude_offset", C.c_float * 3),
        ("exclude_size", C.c_float * 3),
        ("sensor_hrays", C.c_int32),
        ("sensor_vrays", C.c_int32),
        ("sensor_vfov", C.c_float),
        ("lut_directions", C.c_void_p),
        ("lut_offsets", C.c_void_p),
        ("mask", C.c_void_p),
        ("device", C.c_int32),
        ("max_batch_frames", C.c_int32),
    ]


class DynParams(C.Structure):
    _fields_ = [
        ("ground_points_max_distance", C.c_double),
        ("output__position_sigma", C.c_double),
        ("voxel_map__scores__point", C.c_double),
        ("voxel_map__scores__unknown", C.c_double),
        ("voxel_map__scores__ray", C.c_double),
        ("voxel_map__thresholds__apriori_map", C.c_double),
        ("voxel_map__thresholds__new_obstacles", C.c_double),
        ("voxel_map__thresholds__sure_obstacles", C.c_double),
        ("voxel_map__thresholds__frontiers", C.c_double),
        ("classification__min_points", C.c_int32),
        ("classification__max_size", C.c_double),
        ("classification__max_distance", C.c_double),
        ("classification__max_explore_distance", C.c_double),
        ("raycast__pause", C.c_int32),
        ("raycast__new_update_rule", C.c_int32),
        ("raycast__max_distance", C.c_double),
        ("raycast__min_intensity", C.c_double),
        ("raycast__weight_coefficient", C.c_double),
        ("sepclusters__pause", C.c_int32),
        ("sepclusters__max_bg_distance", C.c_double),
        ("sepclusters__min_sure_points", C.c_int32),
    ]


class Scan(C.Structure):
    _fields_ = [
        ("x", C.c_void_p),
        ("y", C.c_void_p),
        ("z", C.c_void_p),
        ("intensity", C.c_void_p),
        ("range", C.c_void_p),
        ("stride_bytes", C.c_size_t),
        ("width", C.c_int32),
        ("height", C.c_int32),
        ("memspace", C.c_int32),
        ("stamp", C.c_double),
    ]


class Detection(C.Structure):
    _fields_ = [
        ("id", C.c_uint32),
        ("frame", C.c_uint32),
        ("n_points", C.c_uint64),
        ("confidence", C.c_double),
        ("detection_probability", C.c_double),
        ("position", C.c_double * 3),
        ("covariance", C.c_double * 9),
    ]


class ClusterInfo(C.Structure):
    _fields_ = [
        ("first_member", C.c_uint32),
        ("n_points", C.c_uint32),
        ("is_close", C.c_int32),
        ("cclass", C.c_int32),
        ("aabb_min", C.c_float * 3),
        ("aabb_max", C.c_float * 3),
        ("obb_center", C.c_float * 3),
        ("obb_size", C.c_float),
    ]


class ScanDebug(C.Structure):
    _fields_ = [
        ("weighted", C.c_void_p),
        ("labels", C.c_void_p),
        ("weighted_cap", C.c_size_t),
        ("n_weighted", C.c_size_t),
        ("clusters", C.c_void_p),
        ("clusters_cap", C.c_size_t),
        ("n_clusters", C.c_size_t),
        ("n_input_after_crop", C.c_uint64),
        ("n_bg_voxels", C.c_uint64),
        ("background_pts_sufficient", C.c_int32),
        ("sure_background_sufficient", C.c_int32),
        ("stage_ms", C.c_double * 8),
        ("far_only", C.c_int32),  # input (batches): the close-first view - far clusters only, LABEL_NONE elsewhere
        ("reserved_", C.c_int32),
    ]


LABEL_NONE = 0xFFFFFFFF


class StatusInfo(C.Structure):
    _fields_ = [
        ("detection_its", C.c_int32),
        ("last_detection_id", C.c_uint32),
        ("background_pts_sufficient", C.c_int32),
        ("sure_background_sufficient", C.c_int32),
        ("raycast_pending", C.c_int32),
        ("map_size", C.c_int32 * 3),
        ("map_offset", C.c_float * 3),
    ]


class CloudView(C.Structure):
    _fields_ = [
        ("x", C.c_void_p),
        ("y", C.c_void_p),
        ("z", C.c_void_p),
        ("intensity", C.c_void_p),
        ("stride_bytes", C.c_size_t),
        ("n", C.c_size_t),
        ("memspace", C.c_int32),
    ]


class GridDesc(C.Structure):
    _fields_ = [
        ("leaf", C.c_float * 3),
        ("offset", C.c_float * 3),
        ("min_b", C.c_int32 * 3),
        ("div_b", C.c_int32 * 3),
    ]


POINT_XYZR = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("range", "<u4")])
DETECTION = np.dtype(
    [
        ("id", "<u4"),
        ("frame", "<u4"),
        ("n_points", "<u8"),
        ("confidence", "<f8"),
        ("detection_probability", "<f8"),
        ("position", "<f8", 3),
        ("covariance", "<f8", 9),
    ]
)
CLUSTER_INFO = np.dtype(
    [
        ("first_member", "<u4"),
        ("n_points", "<u4"),
        ("is_close", "<i4"),
        ("cclass", "<i4"),
        ("aabb_min", "<f4", 3),
        ("aabb_max", "<f4", 3),
        ("obb_center", "<f4", 3),
        ("obb_size", "<f4"),
    ]
)
assert DETECTION.itemsize == C.sizeof(Detection) == 128
assert CLUSTER_INFO.itemsize == C.sizeof(ClusterInfo)
assert POINT_XYZR.itemsize == 16


def declared_entry_points() -> list[str]:
    """Names of every function include/vofod.h declares (without prefix)."""
    text = HEADER.read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\bvofod_(\w+)\s*\(", text)
    seen, out = set(), []
    for n in names:
        if n not in seen:
            seen.add(n)
            out.append(n)
    return out


_P = C.POINTER
_SIGS = {
    "default_params": (None, [_P(StaticParams), _P(DynParams)]),
    "create": (C.c_int, [_P(StaticParams), _P(DynParams), _P(C.c_void_p)]),
    "destroy": (None, [C.c_void_p]),
    "reset": (C.c_int, [C.c_void_p]),
    "set_dynamic_params": (C.c_int, [C.c_void_p, _P(DynParams)]),
    "last_error_string": (C.c_char_p, [C.c_void_p]),
    "get_status": (C.c_int, [C.c_void_p, _P(StatusInfo)]),
    "load_apriori": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "ingest_apriori": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_double, C.c_void_p, _P(C.c_size_t), _P(C.c_size_t)]),
    "ouster_lut": (C.c_int, [C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mask_layout": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    "check_sensor_params": (C.c_int, [_P(Scan), C.c_void_p, C.c_void_p, C.c_void_p, _P(C.c_int32)]),
    "serialize_detections": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, _P(C.c_size_t)]),
    "serialize_status": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, _P(C.c_size_t)]),
    "serialize_profiling_info": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint8, C.c_void_p, C.c_size_t, _P(C.c_size_t)]),
    "comm_unique_id": (C.c_int, [C.c_void_p]),
    "comm_create": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, _P(C.c_void_p)]),
    "comm_destroy": (None, [C.c_void_p]),
    "comm_last_error": (C.c_char_p, [C.c_void_p]),
    "detection_slot_bytes": (C.c_size_t, [C.c_size_t]),
    "pack_detection_slots": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]),
    "unpack_detection_slots": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]),
    "allgather_detections": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]),
    "voxels_as_pc": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_size_t, _P(C.c_size_t)]),
    "update_ground": (C.c_int, [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "read_map": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]),
    "write_map": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]),
    "process_scan": (C.c_int, [C.c_void_p, _P(Scan), C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, _P(C.c_size_t), _P(ScanDebug)]),
    "process_batch": (C.c_int, [C.c_void_p, _P(Scan), C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, _P(C.c_size_t), _P(ScanDebug)]),
    "batch_submit": (C.c_int, [C.c_void_p, _P(Scan), C.c_void_p, C.c_size_t, _P(C.c_int)]),
    "reserve": (C.c_int, [C.c_void_p, C.c_int]),
    "batch_collect": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, _P(C.c_size_t)]),
    "raycast_begin": (C.c_int, [C.c_void_p, _P(Scan), C.c_void_p]),
    "raycast_finish": (C.c_int, [C.c_void_p]),
    "sepclusters_begin": (C.c_int, [C.c_void_p, _P(C.c_int)]),
    "sepclusters_finish": (C.c_int, [C.c_void_p]),
    "voxel_grid_weighted": (C.c_int, [C.c_void_p, _P(CloudView), C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, _P(C.c_size_t), _P(GridDesc)]),
    "voxel_grid_counted": (C.c_int, [C.c_void_p, _P(CloudView), C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_size_t, _P(C.c_size_t), _P(GridDesc)]),
    "cluster": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, _P(GridDesc), C.c_size_t, C.c_float, C.c_void_p, _P(C.c_size_t)]),
    "load_cloud": (C.c_int, [C.c_char_p, C.c_void_p, C.c_size_t, _P(C.c_size_t)]),
    "sim_lut": (C.c_int, [C.c_int32, C.c_int32, C.c_float, C.c_void_p]),
    "profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "profile_read": (C.c_size_t, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
}


# entry points only the product library has to export (include/vofod.h says so)
PRODUCT_ONLY = ("comm_unique_id", "comm_create", "comm_destroy", "comm_last_error", "allgather_detections", "detection_slot_bytes", "pack_detection_slots", "unpack_detection_slots", "serialize_detections", "serialize_status",
                "serialize_profiling_info")


class MsgHeader(C.Structure):
    _fields_ = [("seq", C.c_uint32), ("stamp_sec", C.c_uint32), ("stamp_nsec", C.c_uint32), ("frame_id", C.c_char_p)]


class Library:
    """A loaded implementation of include/vofod.h under a symbol prefix."""

    def __init__(self, path: str | Path, prefix: str = "vofod_"):
        self.path = str(path)
        self.prefix = prefix
        self.cdll = C.CDLL(self.path)
        missing = []
        for name in declared_entry_points():
            sym = prefix + name
            try:
                fn = getattr(self.cdll, sym)
            except AttributeError:
                if name in PRODUCT_ONLY and prefix != "vofod_":
                    continue  # the collective of the batched mode: the CPU oracle has no device to gather on
                missing.append(sym)
                continue
            res, args = _SIGS[name]
            fn.restype = res
            fn.argtypes = args
            setattr(self, name, fn)
        if missing:
            raise ImportError(f"{self.path} does not export: {', '.join(missing)}")

    def extra(self, symbol: str, restype, argtypes):
        fn = getattr(self.cdll, symbol)
        fn.restype = restype
        fn.argtypes = argtypes
        return fn


def ptr(a: np.ndarray | None):
    if a is None:
        return None
    return a.ctypes.data_as(C.c_void_p)
